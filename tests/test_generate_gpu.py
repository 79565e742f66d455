"""Generation parity on the GPU against the real reference's trajectories (golden G9, tools/make_golden.py) and the
pinned CPU oracle: `t2i_generate_ar` (models/unigen.py:457-521), `mmu_generate` (:523-581; KV-cached, recompute and
batched forms) and every round of `t2i_generate` (:344-455; incremental prefix cache on and off).

Integer outputs are compared exactly.  The HIP path computes in bf16 (the reference's autocast mode), so a token is only
required to match while the ORACLE's own top-2 logit margin is above bf16 noise (0.05 on logits of magnitude ~10): after
a near-tie both continuations are legitimate and the comparison of that row stops."""
import math

import pytest
import torch

from helpers import additive, golden, llm_config_dir, oracle_lm

pytestmark = pytest.mark.gpu
MARGIN = 0.05


def _model(g, dev, std=0.02):
    from models import UniGen
    from oracle import weights
    cfg, ids = g["cfg"], g["ids"]
    m = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=llm_config_dir(cfg),
               codebook_size=20, num_vq_tokens=16, load_from_pretrained=True, device=dev, init_seed=1)
    names = [(n, tuple(p.shape)) for n, p in m.llm.named_parameters()]
    m.llm.load_state_dict(weights.synth_llm_state(names, seed=g["weight_seed"], std=std), strict=False)
    return m.eval()


def _agree(got, want, margin, what):
    """tokens equal step by step until the oracle's margin drops below bf16 noise; returns #tokens compared"""
    k = 0
    for i in range(len(want)):
        if margin[i] < MARGIN:
            break
        assert i < len(got) and int(got[i]) == int(want[i]), (what, i, [int(t) for t in got], [int(t) for t in want])
        k += 1
    return k


def test_ar_generation_matches_reference_trajectory(dev):
    """G9-AR: tokens of the REAL reference's t2i_generate_ar (bf16 autocast, CFG 3.0, left-padded prompts, 2-D mask)."""
    g = golden("g9_generate.pt")
    model = _model(g, dev, g["weight_std"])
    ar, tv = g["ar"], g["ids"]["text_vocab"]
    want, margin = ar["bf16"]["tokens"], ar["bf16"]["margin"]
    for use_graph, fused in ((False, True), (True, True), (True, False)):
        model.llm.engine.decode_fused = fused
        got = model.t2i_generate_ar(input_ids=ar["cond"].to(dev), uncond_input_ids=ar["uncond"].to(dev),
                                    attention_mask=ar["attention_mask"].to(dev), guidance_scale=ar["scale"], temperature=1.0,
                                    text_vocab_size=tv, image_token_num_per_image=ar["n"], greedy=True, use_graph=use_graph).cpu()
        compared = sum(_agree(got[b], want[b], margin[b], ("ar", use_graph, fused, b)) for b in range(want.shape[0]))
        print(f"AR vs reference golden (graph={use_graph}, fused={fused}): {compared}/{want.numel()} tokens compared, all equal")
        assert compared >= 24, compared
    model.llm.engine.decode_fused = True


def test_ar_decode_session_is_reused_across_calls_and_never_stale(dev, monkeypatch):
    """Round 5: the captured decode step and its static buffers are kept across `t2i_generate_ar` calls of the same shape
    (Best-of-N calls it once per prompt).  A reused session must give exactly the tokens of a fresh capture -- also when the
    prompt, the key-validity mask or the uniforms differ from the call that captured -- and a different shape or sampling constant
    must not reuse it."""
    g = golden("g9_generate.pt")
    model = _model(g, dev, g["weight_std"])
    eng = model.llm.engine
    ar, tv = g["ar"], g["ids"]["text_vocab"]

    def run(cond, uncond, am, scale, seed=None, greedy=True):
        gen = None if seed is None else torch.Generator(device=dev).manual_seed(seed)
        return model.t2i_generate_ar(input_ids=cond.to(dev), uncond_input_ids=uncond.to(dev), attention_mask=am.to(dev), guidance_scale=scale,
                                     temperature=1.0, text_vocab_size=tv, image_token_num_per_image=ar["n"], greedy=greedy, generator=gen).cpu()
    first = run(ar["cond"], ar["uncond"], ar["attention_mask"], ar["scale"])
    sess = eng._ar_session
    assert sess is not None
    again = run(ar["cond"], ar["uncond"], ar["attention_mask"], ar["scale"])
    assert eng._ar_session["graph"] is sess["graph"] and torch.equal(first, again)
    # other prompts and another padding pattern through the SAME session vs a fresh capture of that call
    cond2, am2 = ar["cond"].flip(0).clone(), ar["attention_mask"].clone()
    am2[:, :3] = 0
    reused = run(cond2, ar["uncond"], am2, ar["scale"])
    assert eng._ar_session["graph"] is sess["graph"]
    monkeypatch.setenv("UNIGEN_AR_GRAPH_CACHE", "0")
    fresh = run(cond2, ar["uncond"], am2, ar["scale"])
    monkeypatch.delenv("UNIGEN_AR_GRAPH_CACHE")
    assert torch.equal(reused, fresh)
    # sampling with uniforms from a seeded generator: the session's uniform buffer is refilled per call
    eng._ar_session = None
    a = run(ar["cond"], ar["uncond"], ar["attention_mask"], ar["scale"], seed=5, greedy=False)
    s2 = eng._ar_session
    b = run(ar["cond"], ar["uncond"], ar["attention_mask"], ar["scale"], seed=6, greedy=False)
    c = run(ar["cond"], ar["uncond"], ar["attention_mask"], ar["scale"], seed=5, greedy=False)
    assert eng._ar_session["graph"] is s2["graph"] and not torch.equal(a, b)
    print(f"sampled tokens equal for equal seeds through a reused session: {(a == c).float().mean().item():.3f} of positions")
    # a sampling constant that is baked into a kernel argument: new session
    run(ar["cond"], ar["uncond"], ar["attention_mask"], ar["scale"] + 1.0)
    assert eng._ar_session["graph"] is not s2["graph"]
    # round 6: the kept session -- KV cache, scratch, graph pool -- is released explicitly or on the way back to training
    model.drop_decode_session()
    assert eng._ar_session is None
    run(ar["cond"], ar["uncond"], ar["attention_mask"], ar["scale"])
    assert eng._ar_session is not None
    model.train()
    assert eng._ar_session is None
    model.eval()


def test_mmu_generate_matches_reference_trajectory(dev):
    """G9-mmu: tokens of the REAL reference's mmu_generate (bf16 autocast, greedy, its mmu mask) from the KV-cached path,
    the recompute path and row 0 of the batched path; early stop on eot_token follows the reference rule."""
    g = golden("g9_generate.pt")
    model = _model(g, dev, g["weight_std"])
    mm = g["mmu"]
    idx, new = mm["idx"].to(dev), mm["max_new_tokens"]
    mask = additive(mm["mask_allow"]).to(dev)
    want, margin = mm["bf16"]["tokens"].tolist(), mm["bf16"]["margin"].tolist()
    for use_cache in (True, False):
        got = model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=new, temperature=0.0, use_cache=use_cache)
        k = _agree(got, want, margin, ("mmu", use_cache))
        print(f"mmu_generate(use_cache={use_cache}) vs reference golden: {k}/{new} tokens compared, all equal")
        assert k >= 8 and len(got) == new
    eot = want[3]
    for use_cache in (True, False):
        stop = [int(t) for t in model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=new, temperature=0.0, eot_token=eot,
                                                   use_cache=use_cache)]
        assert stop == want[:want.index(eot) + 1], (stop, want)


def test_mmu_generate_batch_and_embeddings_match_oracle(dev):
    """Fresh prompts (not in any fixture) against oracle.mmu_generate_ref (pinned by G9): three left-padded rows decoded
    together by mmu_generate_batch, the same rows one at a time through the cached and recompute paths, and a prompt given
    as input_embeddings (the w_und_encoder calling convention, unigen.py:573-575)."""
    from oracle import qwen2_ref
    g = golden("g9_generate.pt")
    model = _model(g, dev, g["weight_std"])
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"], std=g["weight_std"])
    pad = g["ids"]["pad"]
    L, lens, new = 40, [40, 27, 33], 10
    gen = torch.Generator().manual_seed(4)
    rows, masks = [], []
    for n in lens:
        ids = torch.full((L,), pad, dtype=torch.long)
        ids[L - n:] = torch.randint(0, 290, (n,), generator=gen)
        allow = torch.tril(torch.ones(L, L, dtype=torch.bool))
        allow[:, :L - n] = False
        allow[torch.arange(L - n), torch.arange(L - n)] = True      # pad rows attend to themselves (no empty row)
        k = L - n + 9                                               # a bidirectional "image" block like the mmu masks
        allow[L - n:k, L - n:k] = True
        rows.append(ids); masks.append(allow)
    idx, allow = torch.stack(rows), torch.stack(masks)
    mask = additive(allow)
    batch = model.mmu_generate_batch(idx=idx.to(dev), attention_mask=mask.to(dev), max_new_tokens=new, temperature=0.0)
    total = 0
    for r in range(3):
        want, margin = qwen2_ref.mmu_generate_ref(lm, idx=idx[r:r + 1], attention_mask=mask[r:r + 1], max_new_tokens=new, autocast=True)
        total += _agree(batch[r], want, margin, ("batch", r))
        for use_cache in (True, False):
            one = model.mmu_generate(idx=idx[r:r + 1].to(dev), attention_mask=mask[r:r + 1].to(dev), max_new_tokens=new,
                                     temperature=0.0, use_cache=use_cache)
            total += _agree(one, want, margin, ("single", r, use_cache))
    print(f"mmu batch / cached / recompute vs oracle: {total}/{9 * new} tokens compared, all equal")
    assert total >= 45, total
    # embeddings in, embeddings appended per step
    model.register_to_config(w_und_encoder=True)
    try:
        with torch.no_grad():
            emb = lm.model.embed_tokens(idx[:1])
        want, margin = qwen2_ref.mmu_generate_ref(lm, input_embeddings=emb, attention_mask=mask[:1], max_new_tokens=new, autocast=True)
        for use_cache in (True, False):
            got = model.mmu_generate(input_embeddings=emb.to(dev), attention_mask=mask[:1].to(dev), max_new_tokens=new,
                                     temperature=0.0, use_cache=use_cache)
            assert _agree(got, want, margin, ("emb", use_cache)) >= 5
    finally:
        model.register_to_config(w_und_encoder=False)


@pytest.mark.parametrize("incremental", [True, False])
def test_maskgit_every_round_matches_oracle(dev, incremental):
    """All 4 rounds of t2i_generate vs oracle.maskgit_generate_ref driven by the SAME uniforms (InverseCdfSampler is the
    kernel's sampling rule; the oracle with the reference's own torch sampler reproduces the real trajectory, G6).  Per
    image and round: sampled ids must be equal wherever the draw is further than 0.004 from a CDF step, the re-masking
    decision wherever the confidence is further than 0.02 from the threshold; an image leaves the comparison after its
    first legitimate near-tie divergence.  Rounds 2+ exercise the compounded temperature, mask_len clamping and (with
    incremental=True) the cached prefix keys / values."""
    from oracle import qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    model = _model(g, dev)
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    m, ids = g["maskgit"], g["ids"]
    n, N, T = 16, m["input_ids"].shape[0], 4
    sched = lambda t: torch.cos(t * math.pi * 0.5)
    am = additive(m["mask_allow"])
    rounds_checked = 0
    for seed in (77, 78, 79):
        gen = torch.Generator(device=dev).manual_seed(seed)
        state = gen.get_state()
        u = torch.stack([torch.rand((2, N, n), device=dev, generator=gen) for _ in range(T)]).cpu()      # [T, 2, N, n]
        gen.set_state(state)
        trace_hip = []
        got = model.t2i_generate(input_ids=m["input_ids"].to(dev), uncond_input_ids=m["uncond_ids"].to(dev), attention_mask=am.to(dev),
                                 guidance_scale=m["scale"], temperature=1.0, timesteps=T, noise_schedule=sched, generator=gen,
                                 image_token_num_per_image=n, text_vocab_size=ids["text_vocab"], incremental=incremental,
                                 trace=trace_hip).cpu()
        trace = []
        want = qwen2_ref.maskgit_generate_ref(lm, m["input_ids"], m["uncond_ids"], am, m["scale"], 1.0, T, sched, n, ids["text_vocab"],
                                              ids["mask"], qwen2_ref.InverseCdfSampler(u[:, 0], u[:, 1]), autocast=True, trace=trace)
        assert len(trace_hip) == T and got.shape == want.shape
        alive = [True] * N
        for r in range(T):
            probs = trace[r]["mixed"].softmax(-1)
            gap = (probs.cumsum(-1) - u[r, 0].reshape(N, n, 1)).abs().min(-1).values
            s_hip, next_hip = trace_hip[r][0].cpu(), trace_hip[r][1].cpu()
            mk_hip = next_hip == ids["mask"]
            for b in range(N):
                if not alive[b]:
                    continue
                known = trace[r]["sampled"][b] != s_hip[b]
                assert not bool((known & (gap[b] > 0.004)).any()), (seed, r, b, s_hip[b], trace[r]["sampled"][b])
                if bool(known.any()):
                    alive[b] = False
                    continue
                far = (trace[r]["conf"][b] - trace[r]["thr"][b]).abs() > 0.02
                dm = mk_hip[b] != trace[r]["masking"][b]
                assert not bool((dm & far).any()), (seed, r, b, mk_hip[b], trace[r]["masking"][b])
                if bool(dm.any()):
                    alive[b] = False
                    continue
                rounds_checked += 1
                if r == T - 1:
                    assert torch.equal(got[b], want[b])
    print(f"MaskGIT (incremental={incremental}): {rounds_checked}/{3 * N * T} image-rounds followed to exact agreement")
    assert rounds_checked >= 3 * N * T // 2, rounds_checked


def test_maskgit_accepts_the_compressed_mask_from_ids(dev):
    """t2i_generate with the compressed mask `ops.mask_from_ids` builds straight from the ids (what
    bench.py's MaskGIT case passes) takes the same incremental path and returns the same tokens as with the dense additive
    mask of create_attention_mask_predict_next (golden G2's mask)."""
    from unigen_hip import ops
    g = golden("g2_tiny_unigen.pt")
    model = _model(g, dev)
    m, ids = g["maskgit"], g["ids"]
    n = 16
    both = torch.cat([m["input_ids"], m["uncond_ids"]]).to(dev)
    mb = ops.mask_from_ids(both, ids["pad"], ids["soi"], ids["eoi"], ops.MASK_T2I)
    L = both.shape[1]
    cols = torch.arange(mb.nW * 64, device=dev)
    allow = ((mb.bits[:, :, cols // 64] >> (cols % 64)) & 1).bool()[:, :, :L]
    assert torch.equal(allow.cpu(), m["mask_allow"])
    sched = lambda t: torch.cos(t * math.pi * 0.5)
    outs = []
    for mask in (additive(m["mask_allow"]).to(dev), mb):
        outs.append(model.t2i_generate(input_ids=m["input_ids"].to(dev), uncond_input_ids=m["uncond_ids"].to(dev), attention_mask=mask,
                                       guidance_scale=m["scale"], temperature=1.0, timesteps=4, noise_schedule=sched,
                                       generator=torch.Generator(device=dev).manual_seed(5), image_token_num_per_image=n,
                                       text_vocab_size=ids["text_vocab"]).cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("R", [16, 5, 24])
def test_decode_step_forms_agree_at_1p5b_width(dev, monkeypatch, R):
    """`Qwen2Engine.decode_step` at the 1.5B model's width (two layers): the shipped layer (round 6: split-K q/k/v + attention, single-writer
    o / gate-up, down in k-blocks; rows <= 16), the five-launch split-K layer (UNIGEN_DECODE_SW=0, and what 17-32 rows take) and the
    separate-kernel path (`decode_fused = False`) must give the same final-norm hidden state up to bf16 rounding, append the same k / v
    and advance the position; `decode_step_logits` (final norm + head slice in one launch) must match the head applied to that state."""
    from models import UniGen
    from oracle import qwen2_ref, weights
    from unigen_hip import ops
    from unigen_hip.qwen2 import DecodeState
    from helpers import llm_config_dir, rel_err
    cfg = dict(qwen2_ref.QWEN25_1P5B, num_hidden_layers=2, vocab_size=4096)
    model = UniGen(w_und_encoder=False, vocab_size=4096, llm_vocab_size=2048, llm_model_path=llm_config_dir(cfg), codebook_size=2047,
                   num_vq_tokens=16, load_from_pretrained=True, device=dev, init_seed=-1).eval()
    names = [(n, tuple(p.shape)) for n, p in model.llm.named_parameters()]
    model.llm.load_state_dict(weights.synth_llm_state(names, seed=17), strict=False)
    eng = model.llm.engine
    g = torch.Generator().manual_seed(R)
    P, steps = 37, 3
    prompt = (0.02 * torch.randn(R, P, 1536, generator=g)).to(dev)
    xs = [(0.02 * torch.randn(R, 1536, generator=g)).to(dev) for _ in range(steps)]

    def run(fused, sw):
        monkeypatch.setenv("UNIGEN_DECODE_SW", "1" if sw else "0")
        eng.decode_fused = fused
        st = DecodeState(eng.dims, R, P + steps, dev)
        eng.prefill(st, prompt)
        assert eng.decode_sw(st) == (fused and sw and R <= 16)
        hs = [eng.decode_step(st, x.clone()).float().cpu() for x in xs]
        assert int(st.pos.item()) == P + steps and int(st.len.item()) == P + steps + 1
        return hs, [k.float().cpu() for k in st.k], [v.float().cpu() for v in st.v]
    with torch.no_grad():
        wide = run(False, False)
        for fused, sw in ((True, False), (True, True)):
            got = run(fused, sw)
            for i in range(steps):
                e = rel_err(got[0][i], wide[0][i])
                assert e < 1.5e-2, (fused, sw, i, e)
            for a, b in zip(got[1] + got[2], wide[1] + wide[2]):
                assert rel_err(a[:, :, P:], b[:, :, P:]) < 1.5e-2
        if R <= 16:                                     # the head slice in the layer's last launch
            monkeypatch.setenv("UNIGEN_DECODE_SW", "1")
            eng.decode_fused = True
            st = DecodeState(eng.dims, R, P + steps, dev)
            eng.prefill(st, prompt)
            w_head = eng.fp.w("embed")[2048:4095]
            logits = torch.zeros(R, 2047, device=dev)
            eng.decode_step_logits(st, xs[0].clone(), w_head, logits)
            ref = wide[0][0].to(torch.bfloat16).float() @ w_head.float().cpu().t()
            assert rel_err(logits, ref) < 1.5e-2 and int(st.pos.item()) == P + 1
    eng.decode_fused = True
