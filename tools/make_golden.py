"""Generate the golden vectors under tests/golden/ by running the REAL reference (imported from
/root/reference through tools/ref_shims.py) on seeded synthetic inputs, and check the CPU oracle
(oracle/) against it at generation time.  Run in the build container only:

    python tools/make_golden.py            # writes tests/golden/*.pt, prints oracle-vs-reference deltas

The fixtures contain inputs and expected outputs only (no reference source).  Weights are never
stored: they are rebuilt from (name, shape, seed) by oracle/weights.py.
"""
import os
import re
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import ref_shims  # noqa: E402

ref_shims.install()

from oracle import host_ref, magvit_ref, qwen2_ref, weights  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# ------------------------------------------------------------------ a deterministic fake tokenizer
TEXT_V = 300
SPECIALS = ["[PAD]", "<|im_start|>", "<|im_end|>"]


class _Enc(dict):
    """tokenizer output: both enc['input_ids'] and enc.input_ids (prompting_utils.py:67-70,394-420)"""
    __getattr__ = dict.__getitem__


class FakeTok:
    """Covers exactly what UniversalPromptingQwen2 and the step body touch."""

    def __init__(self):
        self.table = {s: TEXT_V + i for i, s in enumerate(SPECIALS)}
        self.pad_token_id = self.table["[PAD]"]
        self.eos_token_id = self.table["<|im_end|>"]
        self.vocab_size = TEXT_V
        self.model_max_length = 4096

    def add_tokens(self, toks):
        for t in toks:
            if t not in self.table:
                self.table[t] = TEXT_V + len(self.table)

    def convert_tokens_to_ids(self, toks):
        return [self.table[t] for t in toks]

    def __len__(self):
        return TEXT_V + len(self.table)

    def _enc(self, s):
        pat = "(" + "|".join(re.escape(k) for k in sorted(self.table, key=len, reverse=True)) + ")"
        ids = []
        for piece in re.split(pat, s):
            if piece in self.table:
                ids.append(self.table[piece])
            else:
                ids += [(ord(c) * 7 + 3) % TEXT_V for c in piece]
        return ids

    def __call__(self, text, **kw):
        ids = [self._enc(t) for t in text] if isinstance(text, (list, tuple)) else self._enc(text)
        return _Enc(input_ids=ids)


def maxdiff(a, b):
    return (a.float() - b.float()).abs().max().item()


# ------------------------------------------------------------------ G4: attention-mask builders
def golden_masks():
    from training.prompting_utils import (create_attention_mask_for_mmu, create_attention_mask_for_mmu_vit,
                                          create_attention_mask_predict_next)
    PAD, SOI, EOI = 300, 303, 304
    g = torch.Generator().manual_seed(11)
    L = 24
    seqs = []
    for npad, ntext, nimg in [(4, 8, 10), (0, 12, 10), (9, 3, 10)]:
        s = [PAD] * npad + torch.randint(0, 290, (ntext,), generator=g).tolist() + [SOI] + \
            torch.randint(312, 330, (nimg,), generator=g).tolist() + [EOI]
        seqs.append(s)
    t2i = torch.tensor(seqs)
    lm = torch.tensor([torch.randint(0, 290, (L - 5,), generator=g).tolist() + [PAD] * 5])
    mmu = torch.tensor([[301, 308, SOI] + torch.randint(312, 330, (8,), generator=g).tolist() + [EOI]
                        + torch.randint(0, 290, (L - 12,), generator=g).tolist()] * 2)
    out = {"t2i_seq": t2i, "lm_seq": lm, "mmu_seq": mmu, "ids": dict(pad=PAD, soi=SOI, eoi=EOI)}
    ref = create_attention_mask_predict_next(t2i, pad_id=PAD, soi_id=SOI, eoi_id=EOI, rm_pad_in_image=True)
    out["t2i_additive"] = ref
    out["t2i_allow"] = (ref[:, 0] == 0)
    mine = host_ref.mask_predict_next_ref(t2i, PAD, SOI, EOI, rm_pad_in_image=True)
    assert torch.equal(mine, out["t2i_allow"]), "oracle t2i mask != reference"
    assert torch.equal(host_ref.to_additive(mine), ref)
    ref = create_attention_mask_predict_next(lm, pad_id=PAD, soi_id=SOI, eoi_id=EOI)
    out["lm_allow"] = (ref[:, 0] == 0)
    assert torch.equal(host_ref.mask_predict_next_ref(lm, PAD, SOI, EOI), out["lm_allow"])
    ref = create_attention_mask_for_mmu(mmu, eoi_id=EOI)
    out["mmu_allow"] = (ref[:, 0] == 0)
    assert torch.equal(host_ref.mask_mmu_ref(mmu, EOI), out["mmu_allow"])
    emb = torch.zeros(2, L, 4)
    ref = create_attention_mask_for_mmu_vit(emb, prefix_length=5, num_tokens=9)
    out["mmu_vit_allow"] = (ref[:, 0] == 0)
    assert torch.equal(host_ref.mask_mmu_vit_ref(2, L, prefix_length=5, num_tokens=9), out["mmu_vit_allow"])
    torch.save(out, os.path.join(OUT, "g4_masks.pt"))
    print("G4 masks: oracle == reference (4 builders)")


# ------------------------------------------------------------------ G5: sampling helpers + masking
def golden_sampling():
    import importlib
    import math
    ref_s = importlib.import_module("models.sampling")
    from data.masking import mask_or_random_replace_tokens
    out = {}
    t = torch.linspace(0, 1, 11)
    out["t"] = t
    for name in ("cosine", "linear", "pow2", "pow0.5", "sigmoid"):
        out["sched_" + name] = ref_s.get_mask_chedule(name)(t)
    g = torch.Generator().manual_seed(5)
    probs = torch.rand(3, 16, generator=g)
    mask_len = torch.tensor([[3], [1], [9]])
    out["probs"], out["mask_len"] = probs, mask_len
    out["topk_mask"] = ref_s.mask_by_random_topk(mask_len, probs, 0.7, generator=torch.Generator().manual_seed(6))
    logits = torch.randn(2, 50, generator=g)
    out["filter_in"] = logits.clone()
    out["filter_k5"] = ref_s.top_k_top_p_filtering(logits.clone(), top_k=5)
    out["filter_p"] = ref_s.top_k_top_p_filtering(logits.clone(), top_p=0.8)
    out["gumbel"] = ref_s.gumbel_noise(torch.zeros(2, 5), generator=torch.Generator().manual_seed(7))
    # MaskGIT training-time masking with the global RNG seeded (data/masking.py draws rand(B) then rand(B,N))
    cfgd = types.SimpleNamespace(training=types.SimpleNamespace(min_masking_rate=0.0, get=lambda k, d=None: d),
                                 model=types.SimpleNamespace(codebook_size=20))
    toks = torch.randint(312, 332, (4, 16), generator=g)
    torch.manual_seed(123)
    ids, labels, _, mp = mask_or_random_replace_tokens(toks, 332, cfgd, mask_schedule=ref_s.cosine_schedule)
    torch.manual_seed(123)
    ts, sc = torch.rand(4), torch.rand(4, 16)
    i2, l2, mp2 = host_ref.maskgit_train_mask_ref(toks, 332, ts, sc, lambda x: torch.cos(x * math.pi * 0.5))
    assert torch.equal(ids, i2) and torch.equal(labels, l2) and torch.equal(mp, mp2), "oracle masking != reference"
    out.update(mask_tokens=toks, mask_ids=ids, mask_labels=labels, mask_prob=mp, mask_seed=123)
    torch.save(out, os.path.join(OUT, "g5_sampling.pt"))
    print("G5 sampling/masking: captured; oracle masking == reference")


# ------------------------------------------------------------------ G2: tiny UniGen step
TINY = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
            rope_theta=1e6, rms_norm_eps=1e-6)


def golden_unigen():
    from models import UniGen
    from training.prompting_utils import (UniversalPromptingQwen2, create_attention_mask_for_mmu,
                                          create_attention_mask_predict_next)
    tok = FakeTok()
    NVQ, CODEBOOK, MAXTXT = 16, 20, 21
    up = UniversalPromptingQwen2(tok, max_seq_len=MAXTXT + NVQ + 3, cond_dropout_prob=0.0, ignore_id=-100)
    V = len(tok) + CODEBOOK + 1
    assert len(tok) == 312 and V == 333
    mask_id = V - 1
    cfg = qwen2_ref.Qwen2Cfg(vocab_size=V, **TINY)
    d = ref_shims.write_llm_config_dir(cfg.to_hf_dict())
    torch.manual_seed(0)
    model = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=len(tok), llm_model_path=d,
                   codebook_size=CODEBOOK, num_vq_tokens=NVQ, load_from_pretrained=True)
    model.train()
    names = [(n, tuple(p.shape)) for n, p in model.llm.named_parameters()]
    sd = weights.synth_llm_state(names, seed=21)
    model.llm.load_state_dict(sd, strict=False)
    assert model.llm.lm_head.weight.data_ptr() == model.llm.model.embed_tokens.weight.data_ptr()

    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, CODEBOOK, (3, NVQ), generator=g) + len(tok)
    ts, sc = torch.rand(2, generator=g), torch.rand(2, NVQ, generator=g)
    import math
    in_img, lab_img, _ = host_ref.maskgit_train_mask_ref(img[:2], mask_id, ts, sc, lambda x: torch.cos(x * math.pi * 0.5))
    def rand_text(n):
        return "".join(chr(97 + v) for v in torch.randint(0, 26, (n,), generator=g).tolist())
    texts_s = [rand_text(5), rand_text(14)]
    texts = tok(texts_s).input_ids
    ids_t2i, _, lab_t2i = up((list(texts_s), in_img, lab_img), 't2i')
    lm_s = [rand_text(17)]
    lm_texts = tok(lm_s).input_ids
    ids_lm, _, lab_lm = up((list(lm_s), ids_t2i.shape[-1]), 'lm')
    ids_mmu, _, lab_mmu = up((img[2:3], [rand_text(9)]), 'mmu')
    PAD, SOI, EOI = int(up.sptids_dict['<|pad|>']), int(up.sptids_dict['<|soi|>']), int(up.sptids_dict['<|eoi|>'])
    m_t2i = create_attention_mask_predict_next(ids_t2i, pad_id=PAD, soi_id=SOI, eoi_id=EOI, rm_pad_in_image=True)
    m_lm = create_attention_mask_predict_next(ids_lm, pad_id=PAD, soi_id=SOI, eoi_id=EOI)
    m_mmu = create_attention_mask_for_mmu(ids_mmu, eoi_id=EOI)
    input_ids = torch.cat([ids_t2i, ids_lm, ids_mmu])
    labels = torch.cat([lab_t2i, lab_lm, lab_mmu])
    mask = torch.cat([m_t2i, m_lm, m_mmu]).to(torch.float32)        # mask_dtype = embedding dtype (train.py:498-503,597)
    kw = dict(batch_size_t2i=2, batch_size_lm=1, batch_size_mmu=1, max_seq_length=MAXTXT, num_vq_tokens=NVQ)

    out = {"cfg": dict(TINY, vocab_size=V), "weight_seed": 21, "input_ids": input_ids, "labels": labels,
           "mask_allow": (mask[:, 0] == 0), "kw": kw, "ids": dict(pad=PAD, soi=SOI, eoi=EOI, mask=mask_id, text_vocab=len(tok)),
           "layout": dict(t2i_texts=texts, t2i_in=in_img, t2i_lab=lab_img, ids_t2i=ids_t2i, lab_t2i=lab_t2i,
                          lm_texts=lm_texts, ids_lm=ids_lm, lab_lm=lab_lm, max_seq_len=up.max_seq_len,
                          conv_start=tok("<|im_start|><|t2i|>user\n").input_ids,
                          conv_end=tok("<|im_end|>\n<|im_start|>assistant\n").input_ids)}

    lm_ref = qwen2_ref.RefCausalLM(cfg)
    lm_ref.load_state_dict(sd, strict=False)

    for mode, ac in (("fp32", False), ("bf16", True)):
        model.zero_grad(set_to_none=True)
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if ac else torch.autocast("cpu", enabled=False)
        with ctx:
            logits, l1, l2, l3 = model(input_ids=input_ids, attention_mask=mask, labels=labels, **kw)
        loss = 1.0 * l1.float() + 0.1 * l2.float() + 1.0 * l3.float()
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in model.llm.named_parameters()}
        # --- the oracle, same inputs
        lm_ref.zero_grad(set_to_none=True)
        lo, r1, r2, r3 = qwen2_ref.unigen_forward_ref(lm_ref, input_ids, mask, labels, autocast=ac, **{k: v for k, v in kw.items() if k != "max_seq_length"})
        (1.0 * r1 + 0.1 * r2 + 1.0 * r3).backward()
        gd = max(maxdiff(grads[n], p.grad) for n, p in lm_ref.named_parameters())
        print(f"G2[{mode}] oracle vs reference: logits {maxdiff(lo, logits):.3e}  losses "
              f"{abs(r1.item()-l1.item()):.2e} {abs(r2.item()-l2.item()):.2e} {abs(r3.item()-l3.item()):.2e}  grads {gd:.3e}")
        assert maxdiff(lo, logits) == 0 and gd == 0, "oracle is not bit-identical to the reference on CPU"
        out[mode] = {"logits": logits.detach().to(torch.bfloat16 if ac else torch.float32),
                     "losses": torch.stack([l1.detach().float(), l2.detach().float(), l3.detach().float()]),
                     "grad_norms": {n: gg.norm().item() for n, gg in grads.items()},
                     "grads_small": {n: gg for n, gg in grads.items() if gg.numel() <= 4096},
                     "grad_embed_rows": grads["model.embed_tokens.weight"][[0, 5, 300, 303, 304, 312, 320, 332]],
                     "grad_q0_rows": grads["model.layers.0.self_attn.q_proj.weight"][:4],
                     "grad_down1_rows": grads["model.layers.1.mlp.down_proj.weight"][:4]}
        if ac:      # one AdamW step exactly like training/train.py:291-330 (decay all but names containing 'bias')
            decay = [p for n, p in model.named_parameters() if "bias" not in n]
            nodecay = [p for n, p in model.named_parameters() if "bias" in n]
            opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}],
                                    lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
            opt.step()
            after = dict(model.llm.named_parameters())
            out[mode]["adamw"] = {"lr": 1e-3, "q0_rows": after["model.layers.0.self_attn.q_proj.weight"][:4].detach().clone(),
                                  "bias": after["model.layers.0.self_attn.q_proj.bias"].detach().clone(),
                                  "norm": after["model.norm.weight"].detach().clone()}
            model.llm.load_state_dict(sd, strict=False)
    # --- G6: MaskGIT trajectory (4 steps, CFG, CPU generator) on the same tiny model, fp32 like the eval scripts
    model.eval()
    gen_ids = ids_t2i.clone()
    gen_ids[:, -(NVQ + 1):-1] = mask_id
    un_ids = gen_ids.clone()
    un_ids[:, :-(NVQ + 2)] = PAD     # "empty prompt" rows
    am = torch.cat([m_t2i, create_attention_mask_predict_next(un_ids, pad_id=PAD, soi_id=SOI, eoi_id=EOI, rm_pad_in_image=True)]).float()
    with torch.no_grad():
        traj = model.t2i_generate(input_ids=gen_ids, uncond_input_ids=un_ids, attention_mask=am, guidance_scale=2.0,
                                  temperature=1.0, timesteps=4, noise_schedule=__import__("models.sampling", fromlist=["x"]).cosine_schedule,
                                  generator=torch.Generator().manual_seed(9), image_token_num_per_image=NVQ,
                                  text_vocab_size=len(tok))
    out["maskgit"] = {"input_ids": gen_ids, "uncond_ids": un_ids, "mask_allow": (am[:, 0] == 0), "seed": 9, "steps": 4,
                      "scale": 2.0, "result": traj}
    torch.save(out, os.path.join(OUT, "g2_tiny_unigen.pt"))
    print("G2/G6 tiny UniGen: captured")


# ------------------------------------------------------------------ G1: MAGVITv2
def golden_magvit():
    from models import MAGVITv2
    torch.manual_seed(0)
    vq = MAGVITv2().eval()
    ref_shapes = [(n, tuple(p.shape)) for n, p in vq.named_parameters()]
    mine = magvit_ref.magvit_param_shapes()
    assert sorted(ref_shapes) == sorted(mine), "oracle parameter inventory != reference"
    sd = weights.synth_magvit_state(ref_shapes, seed=31)
    missing = vq.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("quantize.") for k in missing.missing_keys)
    x = weights.synth_images(2, 256, seed=32)
    with torch.no_grad():
        z = vq.encoder(x)
        idx = vq.get_code(x)
        rec = vq.decode_code(idx)
        z_o = magvit_ref.encode_z_ref(sd, x)
        idx_o = magvit_ref.get_code_ref(sd, x)
        rec_o = magvit_ref.decode_code_ref(sd, idx)
    print(f"G1 oracle vs reference: z {maxdiff(z, z_o):.3e} idx_equal {torch.equal(idx, idx_o)} rec {maxdiff(rec, rec_o):.3e}"
          f"  |z| mean {z.abs().mean():.3f} min {z.abs().min():.2e}")
    assert torch.equal(idx, idx_o)
    out = {"weight_seed": 31, "image_seed": 32, "z": z, "indices": idx, "rec_crop": rec[:, :, 96:160, 96:160].clone(),
           "rec_mean": rec.mean(dim=(2, 3)), "rec_std": rec.std(dim=(2, 3)), "rec_sum": rec.double().sum().item(),
           "n_params": sum(p.numel() for p in vq.parameters())}
    torch.save(out, os.path.join(OUT, "g1_magvit.pt"))
    print("G1 MAGVITv2: captured")


if __name__ == "__main__":
    which = sys.argv[1:] or ["masks", "sampling", "unigen", "magvit"]
    if "masks" in which:
        golden_masks()
    if "sampling" in which:
        golden_sampling()
    if "unigen" in which:
        golden_unigen()
    if "magvit" in which:
        golden_magvit()


# ------------------------------------------------------------------ G7: SigLIP tower (small config)
SIGLIP_SMALL = dict(hidden_size=144, intermediate_size=256, num_hidden_layers=4, num_attention_heads=2, num_channels=3,
                    image_size=56, patch_size=14)


def golden_siglip():
    import importlib
    from oracle import siglip_ref
    ref = importlib.import_module("models.multimodal_encoder.siglip_encoder")
    cfg = ref.SigLipVisionConfig(**SIGLIP_SMALL)
    torch.manual_seed(0)
    m = ref.SigLipVisionModel(cfg).eval()
    shapes = siglip_ref.siglip_param_shapes(144, 256, 4, 3, 14, 56)
    have = {n: tuple(p.shape) for n, p in m.named_parameters() if not n.startswith("vision_model.head")}
    assert dict(shapes) == have, set(dict(shapes)) ^ set(have)
    sd = weights.synth_siglip_state(shapes, seed=41)
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    # what SigLipVisionTower.load_model / forward do (siglip_encoder.py:566-590)
    del m.vision_model.encoder.layers[-1:]
    m.vision_model.head = torch.nn.Identity()
    x = torch.rand(2, 3, 56, 56, generator=torch.Generator().manual_seed(42)) * 2 - 1
    with torch.no_grad():
        out = m(x, output_hidden_states=True).hidden_states[-1]
        mine = siglip_ref.siglip_tower_ref(sd, x, num_layers_total=4, num_heads=2, patch=14)
    print(f"G7 SigLIP oracle vs reference: {maxdiff(out, mine):.3e}")
    assert maxdiff(out, mine) < 1e-5
    torch.save({"cfg": SIGLIP_SMALL, "weight_seed": 41, "image_seed": 42, "out": out}, os.path.join(OUT, "g7_siglip.pt"))
    print("G7 SigLIP: captured")


if __name__ == "__main__" and "siglip" in sys.argv[1:]:
    golden_siglip()


# ------------------------------------------------------------------ G8: DPO log-probs (training/train_dpo.py:51-90)
def golden_dpo():
    mod = ref_shims.import_with_stubs("training.train_dpo")
    g = torch.Generator().manual_seed(8)
    B, L, V, n = 4, 24, 50, 16
    logits = 3.0 * torch.randn(B, L, V, generator=g)
    labels = torch.randint(0, V, (B, L), generator=g)
    labels[torch.rand(B, L, generator=g) < 0.4] = -100
    labels[:, -(n + 1)] = 7                      # at least one kept position per row
    out = {"logits": logits, "labels": labels, "n": n}
    for mode in ("mask", "ar"):
        for avg in (False, True):
            want = mod.get_batch_logps(logits, labels, average_log_prob=avg, num_vq_tokens=n, t2i_gen_mode=mode)
            mine = host_ref.batch_logps_ref(logits, labels, n, average_log_prob=avg, t2i_gen_mode=mode)
            assert torch.equal(want, mine), (mode, avg)
            out[f"{mode}_{int(avg)}"] = want
    torch.save(out, os.path.join(OUT, "g8_dpo_logps.pt"))
    print("G8 DPO log-probs: captured (oracle bit-identical)")


# ------------------------------------------------------------------ G3: one Qwen2.5-1.5B-width decoder layer at L = 387
WIDE = dict(hidden_size=1536, intermediate_size=8960, num_hidden_layers=1, num_attention_heads=12, num_key_value_heads=2,
            rope_theta=1e6, rms_norm_eps=1e-6)


def golden_wide(MAXTXT=128, text_lens=(37, 101), fname="g3_wide_layer.pt", tag="G3"):
    """The real reference UniGen (transformers Qwen2 under it) with ONE decoder layer of the 1.5B model's width on the
    pt1 sequence shape (128 text + 256 image + 3 = 387, left padding), bf16 autocast, forward + backward.  Round 5 (VERDICT r4
    next 6): the same at the benchmarked shape, 512 text + 256 image + 3 = 771 with left padding (`wide771` -> G16)."""
    import math
    from models import UniGen
    from training.prompting_utils import UniversalPromptingQwen2, create_attention_mask_predict_next
    tok = FakeTok()
    NVQ, CODEBOOK = 256, 64
    up = UniversalPromptingQwen2(tok, max_seq_len=MAXTXT + NVQ + 3, cond_dropout_prob=0.0, ignore_id=-100)
    V = len(tok) + CODEBOOK + 1
    mask_id = V - 1
    cfg = qwen2_ref.Qwen2Cfg(vocab_size=V, **WIDE)
    d = ref_shims.write_llm_config_dir(cfg.to_hf_dict())
    torch.manual_seed(0)
    model = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=len(tok), llm_model_path=d, codebook_size=CODEBOOK,
                   num_vq_tokens=NVQ, load_from_pretrained=True)
    model.train()
    names = [(n, tuple(p.shape)) for n, p in model.llm.named_parameters()]
    sd = weights.synth_llm_state(names, seed=33)
    model.llm.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, CODEBOOK, (2, NVQ), generator=g) + len(tok)
    ts, sc = torch.tensor([0.15, 0.8]), torch.rand(2, NVQ, generator=g)
    in_img, lab_img, _ = host_ref.maskgit_train_mask_ref(img, mask_id, ts, sc, lambda x: torch.cos(x * math.pi * 0.5))
    texts = ["".join(chr(97 + v) for v in torch.randint(0, 26, (k,), generator=g).tolist()) for k in text_lens]
    ids, _, labels = up((list(texts), in_img, lab_img), 't2i')
    PAD, SOI, EOI = int(up.sptids_dict['<|pad|>']), int(up.sptids_dict['<|soi|>']), int(up.sptids_dict['<|eoi|>'])
    mask = create_attention_mask_predict_next(ids, pad_id=PAD, soi_id=SOI, eoi_id=EOI, rm_pad_in_image=True).to(torch.float32)
    assert ids.shape == (2, MAXTXT + NVQ + 3) and bool((ids[:, 0] == PAD).all())          # left padding on both rows
    kw = dict(batch_size_t2i=2, batch_size_lm=0, batch_size_mmu=0, max_seq_length=MAXTXT, num_vq_tokens=NVQ)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        logits, l1, _, _ = model(input_ids=ids, attention_mask=mask, labels=labels, **kw)
    l1.float().backward()
    grads = {n: p.grad.detach().clone() for n, p in model.llm.named_parameters()}
    lm_ref = qwen2_ref.RefCausalLM(cfg)
    lm_ref.load_state_dict(sd, strict=False)
    lo, r1, _, _ = qwen2_ref.unigen_forward_ref(lm_ref, ids, mask, labels, autocast=True, batch_size_t2i=2, num_vq_tokens=NVQ)
    r1.backward()
    gd = max(maxdiff(grads[n], p.grad) for n, p in lm_ref.named_parameters())
    print(f"{tag} oracle vs reference: logits {maxdiff(lo, logits):.3e} loss {abs(r1.item() - l1.item()):.2e} grads {gd:.3e}")
    assert maxdiff(lo, logits) == 0 and gd == 0, "oracle is not bit-identical to the reference on CPU"
    out = {"cfg": dict(WIDE, vocab_size=V), "weight_seed": 33, "input_ids": ids, "labels": labels, "mask_allow": (mask[:, 0] == 0),
           "kw": kw, "ids": dict(pad=PAD, soi=SOI, eoi=EOI, mask=mask_id, text_vocab=len(tok)), "codebook": CODEBOOK,
           "loss": l1.detach().float(), "logits_rows": logits.detach()[:, -(NVQ + 1):-1:8].to(torch.bfloat16),
           "grad_norms": {n: gg.norm().item() for n, gg in grads.items()},
           "grad_o_rows": grads["model.layers.0.self_attn.o_proj.weight"][:2].clone(),
           "grad_gate_rows": grads["model.layers.0.mlp.gate_proj.weight"][:2].clone()}
    torch.save(out, os.path.join(OUT, fname))
    print(f"{tag} wide layer: captured (L = {ids.shape[1]}, pads per row {[int((r == PAD).sum()) for r in ids]})")


if __name__ == "__main__" and "dpo" in sys.argv[1:]:
    golden_dpo()
if __name__ == "__main__" and "wide" in sys.argv[1:]:
    golden_wide()
if __name__ == "__main__" and "wide771" in sys.argv[1:]:
    golden_wide(MAXTXT=512, text_lens=(203, 448), fname="g16_wide_layer_L771.pt", tag="G16")


# ------------------------------------------------------------------ G9: greedy generation trajectories (AR image tokens, mmu text)
def golden_generate():
    """The real reference's `t2i_generate_ar` (models/unigen.py:457-521) and `mmu_generate` (:523-581) on the tiny model
    of G2, made deterministic without touching them: AR runs with temperature 1e-6, which turns its
    softmax(logits / temperature) into a one-hot and `torch.multinomial` into argmax; mmu runs with temperature 0 (its own
    argmax branch).  The oracle's ar_generate_ref / mmu_generate_ref must return the same tokens, fp32 and bf16-autocast."""
    from models import UniGen
    g2 = torch.load(os.path.join(OUT, "g2_tiny_unigen.pt"), weights_only=False)
    cfgd, ids = g2["cfg"], g2["ids"]
    V, TV = cfgd["vocab_size"], ids["text_vocab"]
    cfg = qwen2_ref.Qwen2Cfg(**cfgd)
    d = ref_shims.write_llm_config_dir(cfg.to_hf_dict())
    torch.manual_seed(0)
    model = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=TV, llm_model_path=d, codebook_size=20, num_vq_tokens=16,
                   load_from_pretrained=True).eval()
    names = [(n, tuple(p.shape)) for n, p in model.llm.named_parameters()]
    # matrices 7.5x wider than the HF init of G2: with std 0.02 the tied head just echoes the last input token and every
    # trajectory is one repeated id; at 0.15 the layers dominate and the tokens depend on masks, positions and the cache
    STD = float(os.environ.get("G9_STD", "0.15"))
    sd = weights.synth_llm_state(names, seed=g2["weight_seed"], std=STD)
    model.llm.load_state_dict(sd, strict=False)
    lm = qwen2_ref.RefCausalLM(cfg)
    lm.load_state_dict(sd, strict=False)
    out = {"cfg": cfgd, "weight_seed": g2["weight_seed"], "weight_std": STD, "ids": ids}

    # ---- AR image tokens with CFG: left-padded cond / uncond prompts, 2-D attention mask (what inference_t2i.py passes)
    n, B, P = 16, 2, 30
    gen = torch.Generator().manual_seed(5)
    cond = torch.randint(0, 290, (B, P + n + 1), generator=gen)
    uncond = torch.randint(0, 290, (B, P + n + 1), generator=gen)
    cond[0, :6] = ids["pad"]
    uncond[:, :20] = ids["pad"]
    am = torch.cat([cond != ids["pad"], uncond != ids["pad"]]).long()
    am[:, P:] = 1
    out["ar"] = {"cond": cond, "uncond": uncond, "attention_mask": am, "n": n, "P": P, "scale": 3.0}
    for mode, ac in (("fp32", False), ("bf16", True)):
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if ac else torch.autocast("cpu", enabled=False)
        with torch.no_grad(), ctx:
            ce, ue = model.llm.model.embed_tokens(cond), model.llm.model.embed_tokens(uncond)
            ref_tok = model.t2i_generate_ar(input_ids=cond, uncond_input_ids=uncond, input_embeddings=ce, uncond_input_embeddings=ue,
                                            attention_mask=am, guidance_scale=3.0, temperature=1e-6, text_vocab_size=TV,
                                            image_token_num_per_image=n)
        with torch.no_grad():
            mine, margin = qwen2_ref.ar_generate_ref(lm, lm.model.embed_tokens(cond[:, :P]), lm.model.embed_tokens(uncond[:, :P]), n, 3.0,
                                                     TV, key_valid=am[:, :P], autocast=ac)
        same = torch.equal(ref_tok.long(), mine.long())
        print(f"G9 AR[{mode}] reference tokens {ref_tok.tolist()}  oracle equal: {same}  min margin {margin.min():.3f}")
        assert same, "oracle ar_generate_ref != reference t2i_generate_ar"
        out["ar"][mode] = {"tokens": ref_tok.long(), "margin": margin}

    # ---- mmu text continuation: the mmu row of G2 (image tokens then text) under its create_attention_mask_for_mmu mask
    Pm, new = 30, 12
    idx = g2["input_ids"][-1:, :Pm]
    allow = g2["mask_allow"][-1:, :Pm, :Pm]
    mask = host_ref.to_additive(allow).to(torch.float32)
    out["mmu"] = {"idx": idx, "mask_allow": allow, "max_new_tokens": new}
    for mode, ac in (("fp32", False), ("bf16", True)):
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if ac else torch.autocast("cpu", enabled=False)
        with ctx:
            ref_tok = [int(t) for t in model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=new, temperature=0.0)]
        mine, margin = qwen2_ref.mmu_generate_ref(lm, idx=idx, attention_mask=mask, max_new_tokens=new, autocast=ac)
        print(f"G9 mmu[{mode}] reference tokens {ref_tok}  oracle equal: {ref_tok == mine}  min margin {min(margin):.3f}")
        assert ref_tok == mine, "oracle mmu_generate_ref != reference mmu_generate"
        out["mmu"][mode] = {"tokens": torch.tensor(ref_tok), "margin": torch.tensor(margin)}
        # early stop on eot_token: same rule in both
        eot = ref_tok[3]
        with ctx:
            ref_stop = [int(t) for t in model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=new, temperature=0.0, eot_token=eot)]
        stop, _ = qwen2_ref.mmu_generate_ref(lm, idx=idx, attention_mask=mask, max_new_tokens=new, eot_token=eot, autocast=ac)
        assert ref_stop == stop and stop == ref_tok[:ref_tok.index(eot) + 1]
    torch.save(out, os.path.join(OUT, "g9_generate.pt"))
    print("G9 generation trajectories: captured (oracle == reference)")


if __name__ == "__main__" and "generate" in sys.argv[1:]:
    golden_generate()


# ------------------------------------------------------------------ G10: checkpoints written by the reference's own save_pretrained
CKPT_TINY = dict(hidden_size=128, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1,
                 rope_theta=1e6, rms_norm_eps=1e-6)


def golden_checkpoint():
    """The real reference `ModelMixin.save_pretrained` (models/modeling_utils.py:257-399; the call of
    utils/checkpoint.py:53-59 uses safe_serialization=False) writes a very small UniGen twice: one `pytorch_model.bin`
    and a sharded set (`max_shard_size` small) with its index file.  The directories are committed as DATA under
    tests/golden/ and loaded through the build's from_pretrained in tests/test_checkpoint_gpu.py.  safetensors output
    of a tied-embedding UniGen is refused by safetensors itself in the reference, so only the .bin formats exist."""
    import shutil
    from models import UniGen
    V, TV = 333, 312
    cfg = qwen2_ref.Qwen2Cfg(vocab_size=V, **CKPT_TINY)
    d = ref_shims.write_llm_config_dir(cfg.to_hf_dict())
    torch.manual_seed(0)
    model = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=TV, llm_model_path=d, codebook_size=20, num_vq_tokens=16,
                   load_from_pretrained=True).eval()
    names = [(n, tuple(p.shape)) for n, p in model.llm.named_parameters()]
    sd = weights.synth_llm_state(names, seed=55)
    model.llm.load_state_dict(sd, strict=False)
    model.register_to_config(llm_model_path="qwen2.5-ckpt-tiny")        # a name, not this container's temp dir
    ids = torch.randint(0, 290, (2, 24), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        logits = model(input_ids=ids, attention_mask=None)
    for name, kw in (("ckpt_ref_single", {}), ("ckpt_ref_sharded", {"max_shard_size": "200KB"})):
        out = os.path.join(OUT, name)
        shutil.rmtree(out, ignore_errors=True)
        model.save_pretrained(out, safe_serialization=False, **kw)
        print(f"G10 {name}: {sorted(os.listdir(out))}")
    torch.save({"cfg": dict(CKPT_TINY, vocab_size=V), "weight_seed": 55, "input_ids": ids, "logits_last": logits[:, -3:].clone(),
                "keys": sorted(model.state_dict().keys())}, os.path.join(OUT, "g10_checkpoint.pt"))
    print("G10 reference-written checkpoints: captured")


if __name__ == "__main__" and "checkpoint" in sys.argv[1:]:
    golden_checkpoint()


# ------------------------------------------------------------------ G11: prepare_inputs_for_mmu and generate (models/unigen.py:133-228, 584-588)
def golden_mmu_inputs():
    """The real reference UniGen with an mm_projector: `prepare_inputs_for_mmu` in training and eval mode on ragged image
    features (with and without a system prompt), and `generate` (transformers' GenerationMixin underneath) greedy on
    left-padded ids and on embeddings."""
    from models import UniGen
    from training.prompting_utils import UniversalPromptingQwen2
    g2 = torch.load(os.path.join(OUT, "g2_tiny_unigen.pt"), weights_only=False)
    cfgd, ids = g2["cfg"], g2["ids"]
    V, TV, MMD = cfgd["vocab_size"], ids["text_vocab"], 32
    tok = FakeTok()
    up = UniversalPromptingQwen2(tok, max_seq_len=32, cond_dropout_prob=0.0, ignore_id=-100)
    cfg = qwen2_ref.Qwen2Cfg(**cfgd)
    d = ref_shims.write_llm_config_dir(cfg.to_hf_dict())
    torch.manual_seed(0)
    model = UniGen(w_und_encoder=True, vocab_size=V, llm_vocab_size=TV, llm_model_path=d, codebook_size=20, num_vq_tokens=16,
                   load_from_pretrained=True, mm_input_dim=MMD, und_proj_depth=2)
    names = [(n, tuple(p.shape)) for n, p in model.named_parameters() if not n.startswith("llm.lm_head")]
    STD = 0.15
    sd = weights.synth_llm_state(names, seed=61, std=STD)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res
    out = {"cfg": cfgd, "weight_seed": 61, "weight_std": STD, "mm_input_dim": MMD, "ids": ids,
           "template": dict(max_seq_len=32, ignore_id=-100, eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id,
                            task_token_first=False, sptids={k: int(v) for k, v in up.sptids_dict.items()})}
    g = torch.Generator().manual_seed(13)
    B, N, Lt = 3, 12, 14
    feats = torch.randn(B, N, MMD, generator=g)
    shapes = torch.tensor([[3, 4], [2, 3], [2, 5]])
    txt = torch.randint(0, 290, (B, Lt), generator=g)
    txt[0, 9] = tok.eos_token_id; txt[0, 10:] = tok.pad_token_id          # row 0: answer ends early, right-padded
    txt[1, -1] = tok.eos_token_id
    labels = txt.clone(); labels[:, :4] = -100
    sysids = torch.randint(0, 290, (B, 5), generator=g)
    out["mmu_in"] = dict(image_feats=feats, spatial_shapes=shapes, input_ids=txt, label_ids=labels, input_ids_system=sysids)
    for mode in ("train", "eval"):
        model.train(mode == "train")
        for tag, sys_ in (("nosys", None), ("sys", sysids)):
            with torch.no_grad():
                e, am, lab, p1 = model.prepare_inputs_for_mmu(feats, shapes, txt, labels, up, sys_)
            out[f"mmu_{mode}_{tag}"] = dict(embeddings=e, attention_mask=am, labels=lab, part1=p1)
            print(f"G11 prepare_inputs_for_mmu[{mode},{tag}]: emb {tuple(e.shape)} mask {tuple(am.shape)} labels {tuple(lab.shape)}")
    # ---- generate(): greedy, left-padded ids with a 2-D mask; then the same prompts as embeddings
    model.eval()
    P, new = 18, 10
    prompt = torch.randint(0, 290, (2, P), generator=g)
    prompt[1, :5] = tok.pad_token_id
    am = (prompt != tok.pad_token_id).long()
    with torch.no_grad():
        full = model.generate(input_ids=prompt, attention_mask=am, max_new_tokens=new, do_sample=False, use_cache=True,
                              pad_token_id=tok.eos_token_id)
        cont = model.generate(input_embeddings=model.llm.model.embed_tokens(prompt), attention_mask=am, max_new_tokens=new,
                              do_sample=False, use_cache=True, pad_token_id=tok.eos_token_id)
        eos = int(full[0, P + 3])
        stop = model.generate(input_ids=prompt, attention_mask=am, max_new_tokens=new, do_sample=False, use_cache=True,
                              pad_token_id=tok.pad_token_id, eos_token_id=eos)
    assert torch.equal(full[:, :P], prompt) and torch.equal(full[:, P:], cont)
    # margins from the oracle (causal + key-validity mask, positions = arange like UniGen.forward; HF generate numbers the
    # positions from the first real token, which RoPE's relative form makes equivalent up to rounding)
    lm = qwen2_ref.RefCausalLM(cfg)
    lm.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("llm.")}, strict=False)
    toks, margins = [], []
    for r in range(2):
        allow = torch.tril(torch.ones(P, P, dtype=torch.bool)) & am[r].bool()[None, :]
        allow |= torch.eye(P, dtype=torch.bool)
        t, m = qwen2_ref.mmu_generate_ref(lm, idx=prompt[r:r + 1], attention_mask=host_ref.to_additive(allow[None]).float(),
                                          max_new_tokens=new, autocast=False)
        toks.append(t); margins.append(m)
    print(f"G11 generate: reference {full[:, P:].tolist()}\n             oracle    {toks}  stop@{eos}: {stop.tolist()}")
    assert toks == full[:, P:].tolist(), "oracle causal greedy decode != reference generate"
    out["generate"] = dict(prompt=prompt, attention_mask=am, max_new_tokens=new, full=full, cont=cont, eos=eos, stop=stop,
                           margin=torch.tensor(margins))
    torch.save(out, os.path.join(OUT, "g11_mmu_inputs.pt"))
    print("G11 prepare_inputs_for_mmu / generate: captured")


if __name__ == "__main__" and "mmu_inputs" in sys.argv[1:]:
    golden_mmu_inputs()


# ------------------------------------------------------------------ G12: gen_projector path (models/unigen.py:74-92,255-270)
def golden_gen_head():
    """The real reference UniGen built with gen_proj_depth = 2 (both use_gen_dim settings): mixed t2i + lm + mmu batch,
    forward + backward under bf16 autocast; the oracle (unigen_forward_gen_ref) must be bit-identical."""
    from models import UniGen
    g2 = torch.load(os.path.join(OUT, "g2_tiny_unigen.pt"), weights_only=False)
    cfgd, ids = g2["cfg"], g2["ids"]
    V, TV, CB, n = cfgd["vocab_size"], ids["text_vocab"], 20, 16
    cfg = qwen2_ref.Qwen2Cfg(**cfgd)
    d = ref_shims.write_llm_config_dir(cfg.to_hf_dict())
    out = {"cfg": cfgd, "weight_seed": 77, "ids": ids, "codebook": CB, "n": n}
    # the G2 batch with the image slots rewritten as RAW codes (mask id = codebook_size) and raw-code labels
    inp, lab = g2["input_ids"].clone(), g2["labels"].clone()
    bt = g2["kw"]["batch_size_t2i"]
    slot = inp[:bt, -(n + 1):-1]
    inp[:bt, -(n + 1):-1] = torch.where(slot == ids["mask"], CB, slot - TV)
    ls = lab[:bt, -(n + 1):-1]
    lab[:bt, -(n + 1):-1] = torch.where(ls == -100, -100, ls - TV)
    # t2i rows only: the reference applies gen_embed to the last n+1 slots of EVERY row of the batch (:258-259), so a row
    # whose slots hold text ids (lm / mmu rows) indexes past its codebook_size + 1 table -- the path only runs on pure t2i batches
    inp, lab, allow = inp[:bt], lab[:bt], g2["mask_allow"][:bt]
    mask = host_ref.to_additive(allow).float()
    kw = dict(g2["kw"], batch_size_lm=0, batch_size_mmu=0)
    out.update(input_ids=inp, labels=lab, mask_allow=allow, kw=kw)
    for use_dim in (False, True):
        torch.manual_seed(0)
        model = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=TV, llm_model_path=d, codebook_size=CB, num_vq_tokens=n,
                       load_from_pretrained=True, gen_proj_depth=2, use_gen_dim=use_dim, gen_input_dim=16).train()
        assert model.config.mask_token_id == CB
        names = [(k, tuple(p.shape)) for k, p in model.named_parameters()]
        sd = weights.synth_llm_state(names, seed=77, std=0.05)
        res = model.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys
        with torch.autocast("cpu", dtype=torch.bfloat16):
            img_logits, l1, l2, l3 = model(input_ids=inp, attention_mask=mask, labels=lab, **kw)
        l1.float().backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        lm = qwen2_ref.RefCausalLM(cfg)
        lm.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("llm.")}, strict=False)
        gen = qwen2_ref.GenHeadRef(cfgd["hidden_size"], CB, depth=2, use_gen_dim=use_dim, gen_input_dim=16)
        gen.load_state_dict({k: v for k, v in sd.items() if not k.startswith("llm.")})
        okw = {k: v for k, v in kw.items() if k != "max_seq_length"}
        o_logits, r1, r2, r3 = qwen2_ref.unigen_forward_gen_ref(lm, gen, inp, mask, lab, autocast=True, **okw)
        r1.backward()
        gd = max(maxdiff(grads[k], p.grad) for k, p in list(gen.named_parameters()) + [("llm." + k, p) for k, p in lm.named_parameters()]
                 if k in grads)
        print(f"G12[use_gen_dim={use_dim}] oracle vs reference: img_logits {maxdiff(o_logits, img_logits):.3e} losses "
              f"{abs(r1.item() - l1.item()):.1e} grads {gd:.3e}")
        assert maxdiff(o_logits, img_logits) == 0 and gd == 0, "oracle gen branch is not bit-identical to the reference"
        out[f"dim{int(use_dim)}"] = {"img_logits": img_logits.detach().to(torch.bfloat16),
                                     "loss": l1.detach().float(),
                                     "grads_gen_head": {k: (v if v.numel() <= 8192 else v[:4].clone()) for k, v in grads.items()
                                                        if not k.startswith("llm.")},
                                     "grad_norms_gen": {k: v.norm().item() for k, v in grads.items() if not k.startswith("llm.")},
                                     "grad_norms_llm": {k: v.norm().item() for k, v in grads.items() if k.startswith("llm.")}}
    torch.save(out, os.path.join(OUT, "g12_gen_head.pt"))
    print("G12 gen_projector path: captured")


if __name__ == "__main__" and "gen_head" in sys.argv[1:]:
    golden_gen_head()


# ------------------------------------------------------------------ G13: AR image tokens on the gen_projector path (models/unigen.py:486-495,512-514)
def golden_ar_gen_head():
    """The real reference's `t2i_generate_ar` on a model built with gen_proj_depth = 2 (img_head on the last hidden state of
    `llm.model`, gen_projector(gen_embed(token)) as the next input), deterministic through temperature 1e-6 as in G9; both
    use_gen_dim settings, fp32 and bf16-autocast.  The oracle's ar_generate_ref(gen=...) must return the same tokens.
    Under autocast the reference mixes cond / uncond logits in bf16, where the two largest values can TIE (its multinomial then
    draws among them from the global RNG): the prompt seed is the first one whose trajectories have no tie in any mode."""
    from models import UniGen
    g2 = torch.load(os.path.join(OUT, "g2_tiny_unigen.pt"), weights_only=False)
    cfgd, ids = g2["cfg"], g2["ids"]
    V, TV, CB, n, B, P = cfgd["vocab_size"], ids["text_vocab"], 20, 16, 2, 30
    cfg = qwen2_ref.Qwen2Cfg(**cfgd)
    d = ref_shims.write_llm_config_dir(cfg.to_hf_dict())
    STD = 0.15
    built = {}
    for use_dim in (False, True):
        torch.manual_seed(0)
        model = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=TV, llm_model_path=d, codebook_size=CB, num_vq_tokens=n,
                       load_from_pretrained=True, gen_proj_depth=2, use_gen_dim=use_dim, gen_input_dim=16).eval()
        names = [(k, tuple(p.shape)) for k, p in model.named_parameters()]
        sd = weights.synth_llm_state(names, seed=78, std=STD)
        assert not model.load_state_dict(sd, strict=False).unexpected_keys
        lm = qwen2_ref.RefCausalLM(cfg)
        lm.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("llm.")}, strict=False)
        gen = qwen2_ref.GenHeadRef(cfgd["hidden_size"], CB, depth=2, use_gen_dim=use_dim, gen_input_dim=16)
        gen.load_state_dict({k: v for k, v in sd.items() if not k.startswith("llm.")})
        built[use_dim] = (model, lm, gen)
    for prompt_seed in range(6, 60):
        gen_ = torch.Generator().manual_seed(prompt_seed)
        cond = torch.randint(0, 290, (B, P + n + 1), generator=gen_)
        uncond = torch.randint(0, 290, (B, P + n + 1), generator=gen_)
        cond[1, :9] = ids["pad"]
        uncond[:, :18] = ids["pad"]
        am = torch.cat([cond != ids["pad"], uncond != ids["pad"]]).long()
        am[:, P:] = 1
        out = {"cfg": cfgd, "weight_seed": 78, "weight_std": STD, "ids": ids, "codebook": CB, "n": n, "P": P, "scale": 2.5,
               "cond": cond, "uncond": uncond, "attention_mask": am, "prompt_seed": prompt_seed}
        tie = False
        for use_dim in (False, True):
            model, lm, gen = built[use_dim]
            rec = {}
            for mode, ac in (("fp32", False), ("bf16", True)):
                with torch.no_grad():
                    mine, margin = qwen2_ref.ar_generate_ref(lm, lm.model.embed_tokens(cond[:, :P]), lm.model.embed_tokens(uncond[:, :P]), n,
                                                             2.5, TV, key_valid=am[:, :P], autocast=ac, gen=gen)
                if margin.min() <= 0:
                    tie = True
                    break
                ctx = torch.autocast("cpu", dtype=torch.bfloat16) if ac else torch.autocast("cpu", enabled=False)
                with torch.no_grad(), ctx:
                    ce, ue = model.llm.model.embed_tokens(cond), model.llm.model.embed_tokens(uncond)
                    ref_tok = model.t2i_generate_ar(input_ids=cond, uncond_input_ids=uncond, input_embeddings=ce, uncond_input_embeddings=ue,
                                                    attention_mask=am, guidance_scale=2.5, temperature=1e-6, text_vocab_size=TV,
                                                    image_token_num_per_image=n)
                same = torch.equal(ref_tok.long(), mine.long())
                print(f"G13 AR gen head[prompt seed {prompt_seed}, use_gen_dim={use_dim}, {mode}] reference tokens {ref_tok.tolist()}  "
                      f"oracle equal: {same}  min margin {margin.min():.3f}")
                assert same, "oracle ar_generate_ref(gen=...) != reference t2i_generate_ar on the gen_projector path"
                rec[mode] = {"tokens": ref_tok.long(), "margin": margin}
            if tie:
                break
            out[f"dim{int(use_dim)}"] = rec
        if not tie:
            torch.save(out, os.path.join(OUT, "g13_ar_gen_head.pt"))
            print("G13 AR generation on the gen_projector path: captured (oracle == reference)")
            return
        print(f"G13: prompt seed {prompt_seed} has a bf16 tie, trying the next one")
    raise RuntimeError("G13: no tie-free prompt seed found")


if __name__ == "__main__" and "ar_gen_head" in sys.argv[1:]:
    golden_ar_gen_head()


# ------------------------------------------------------------------ G14: RoPE scaling arguments (models/unigen.py:38-40,61-64)
def golden_rope_scaling():
    """The real reference UniGen built with scaling_factor = 2.0 and rope_type 'linear' / 'dynamic' (config.rope_scaling =
    {"factor", "type"}; the tiny config says max_position_embeddings = 16 so that the 40-token batch of G2 is past it and the
    dynamic-NTK base applies): logits and the three losses under bf16 autocast and in fp32; the oracle must be bit-identical."""
    from models import UniGen
    import transformers
    # Third-party API shim (the reference pins transformers 4.51.0, this container has 5.x): in 4.51 `config.rope_scaling =
    # {"factor", "type"}` (models/unigen.py:64) sits NEXT to config.rope_theta; in 5.x `rope_scaling` aliases the whole
    # `rope_parameters` dict, so the same assignment drops rope_theta / rope_type and Qwen2RotaryEmbedding raises.  The shim
    # turns the legacy assignment into the 5.x form of the same request; the rotary arithmetic itself is transformers'.
    cfg_cls = transformers.Qwen2Config
    orig_setattr = cfg_cls.__setattr__

    def legacy_rope_scaling(self, key, value):
        if key == "rope_scaling" and isinstance(value, dict) and "type" in value and "rope_type" not in value:
            theta = (self.__dict__.get("rope_parameters") or {}).get("rope_theta", 10000.0)
            value = {"rope_type": value["type"], "factor": value["factor"], "rope_theta": theta}
            key = "rope_parameters"
        orig_setattr(self, key, value)
    cfg_cls.__setattr__ = legacy_rope_scaling
    g2 = torch.load(os.path.join(OUT, "g2_tiny_unigen.pt"), weights_only=False)
    cfgd, ids = dict(g2["cfg"]), g2["ids"]
    V, TV = cfgd["vocab_size"], ids["text_vocab"]
    mask = host_ref.to_additive(g2["mask_allow"]).float()
    kw = {k: v for k, v in g2["kw"].items() if k != "max_seq_length"}
    out = {"cfg": cfgd, "weight_seed": g2["weight_seed"], "ids": ids, "factor": 2.0, "max_position_embeddings": 16}
    for kind in ("linear", "dynamic"):
        hf = qwen2_ref.Qwen2Cfg(**cfgd).to_hf_dict()
        hf["max_position_embeddings"] = 16
        d = ref_shims.write_llm_config_dir(hf)
        torch.manual_seed(0)
        model = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=TV, llm_model_path=d, codebook_size=20, num_vq_tokens=16,
                       load_from_pretrained=True, scaling_factor=2.0, rope_type=kind).eval()
        names = [(n, tuple(p.shape)) for n, p in model.llm.named_parameters()]
        sd = weights.synth_llm_state(names, seed=g2["weight_seed"])
        model.llm.load_state_dict(sd, strict=False)
        lm = qwen2_ref.RefCausalLM(qwen2_ref.Qwen2Cfg(**cfgd, rope_scaling={"factor": 2.0, "type": kind}, max_position_embeddings=16))
        lm.load_state_dict(sd, strict=False)
        rec = {}
        for mode, ac in (("fp32", False), ("bf16", True)):
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if ac else torch.autocast("cpu", enabled=False)
            with torch.no_grad(), ctx:
                logits, l1, l2, l3 = model(input_ids=g2["input_ids"], attention_mask=mask, labels=g2["labels"], **g2["kw"])
            with torch.no_grad():
                lo, r1, r2, r3 = qwen2_ref.unigen_forward_ref(lm, g2["input_ids"], mask, g2["labels"], autocast=ac, **kw)
            dl = maxdiff(lo, logits.float())
            print(f"G14 rope_type={kind} [{mode}] oracle vs reference: logits {dl:.3e} losses {abs(float(l1) - float(r1)):.1e}")
            assert dl == 0, "oracle rope scaling is not bit-identical to the reference"
            rec[mode] = {"logits": logits.detach().to(torch.bfloat16 if ac else torch.float32),
                         "losses": torch.stack([l1.float(), l2.float(), l3.float()]).detach()}
        # the scaling must change something: compare with the unscaled G2 logits
        assert maxdiff(rec["fp32"]["logits"], g2["fp32"]["logits"]) > 1e-3
        out[kind] = rec
    cfg_cls.__setattr__ = orig_setattr
    torch.save(out, os.path.join(OUT, "g14_rope_scaling.pt"))
    print("G14 rope scaling: captured (oracle == reference)")


if __name__ == "__main__" and "rope_scaling" in sys.argv[1:]:
    golden_rope_scaling()


# ------------------------------------------------------------------ G15: optional branches of data/masking.py (:20-22, :33-66)
def golden_masking_options():
    """The real reference's mask_or_random_replace_tokens at evaluation time with `eval_mask_ratios` and
    `mask_contiguous_region_prob = 1`: every draw comes from Python's `random` (ratios, rectangle bounds), none from torch, so a
    seeded call is reproducible on any device."""
    import random
    from data.masking import mask_or_random_replace_tokens as ref_mask
    out = {"cases": []}
    for n, B, seed in ((256, 6, 7), (16, 4, 11), (1024, 3, 5)):
        cfg = types.SimpleNamespace(training=_Cfg(min_masking_rate=0.0, eval_mask_ratios=[0.25, 0.5, 0.9], mask_contiguous_region_prob=1.0),
                                    model=types.SimpleNamespace(codebook_size=8192))
        g = torch.Generator().manual_seed(seed)
        toks = torch.randint(151674, 151674 + 8192, (B, n), generator=g)
        random.seed(seed)
        ids, labels, lw, mp = ref_mask(toks, 159866, cfg, lambda t: torch.cos(t * 3.141592653589793 * 0.5), is_train=False)
        assert lw is None
        out["cases"].append({"n": n, "B": B, "seed": seed, "tokens": toks, "input_ids": ids, "labels": labels, "mask_prob": mp})
        print(f"G15 n={n}: masked per row {[int(v) for v in (labels != -100).sum(1)]} ratios {mp.tolist()}")
    torch.save(out, os.path.join(OUT, "g15_masking_options.pt"))
    print("G15 masking options: captured")


class _Cfg(dict):
    """OmegaConf-like node: attribute access + .get (what data/masking.py touches)"""
    __getattr__ = dict.__getitem__


if __name__ == "__main__" and "masking_options" in sys.argv[1:]:
    golden_masking_options()
