#!/usr/bin/env python
"""bench.py -- UniGen-1.5B stage-1 (t2i) training step on MI355X, the metric BASELINE.json names:
train-step samples/s for the 1.5B model with 256x256 images (256 image tokens + 512 text tokens,
L = 771), bf16 compute, one process per GPU, gradients all-reduced over RCCL.

One timed "step" = the whole hot path on one synthetic batch already resident in HBM:
  MAGVITv2.get_code(images) -> token layout + dense additive mask (caller-side glue, on device)
  -> UniGen.forward (28-layer backbone, tied lm_head + CE on the 256 image positions)
  -> backward -> flat-gradient all-reduce (N > 1) -> fused AdamW -> zero_grad.
Nothing is skipped, cached across steps or run at reduced size.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: bench.py starts torch.distributed.run itself as a CHILD process before
                                       anything touches the GPU and relays its output and exit code)

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the bf16 MFMA GEMM): algorithmic
flops per launch / average launch duration, measured with HIP events on the launch stream during
instrumented steps that follow the timed ones.  `cpu_baseline` times the CPU oracle (oracle/, kind
"port") on a bounded sample on the host cores; it is a reported baseline, not what is optimised.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool (set before HIP initialises)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# ---- workload constants (SURVEY.md §8 "model constants" / §8d)
TEXT_VOCAB = 151674                 # Qwen2.5 tokenizer (151665) + 9 UniGen specials
CODEBOOK = 8192
VOCAB = TEXT_VOCAB + CODEBOOK + 1   # 159867
MASK_ID = VOCAB - 1
NVQ = 256
PAD, IM_START, SOI, EOI, T2I = 151643, 151644, 151665, 151666, 151669
SEED = 10084                        # configs/unigen_1_5b/unigen_pt1.yaml:92

# algorithmic FLOPs: per t2i sample at L = 771
FLOP_LLM_FWD = 2.2029e12            # linear 2.0203 + attention 0.0569 + lm_head on 256 label rows 0.1257
FLOP_SAMPLE = 3 * FLOP_LLM_FWD + 0.3551e12     # fwd+bwd + MAGVITv2 encode (fwd only) = 6.964 TFLOP
PEAK_BF16 = 2.5e15


def t2i_rows(ops, text, image_in, image_labels):
    """Caller-side glue on the device, through the product's own kernels (both are pinned against the
    real reference: `ug_t2i_assemble` against UniversalPromptingQwen2.t2i_prompt's ids / labels, golden G2 layout, and
    `ug_attn_mask_from_ids` against create_attention_mask_predict_next(rm_pad_in_image=True), golden G4):
      [<|im_start|><|t2i|> text ... | <|soi|> 256 image ids <|eoi|>] (training/prompting_utils.py:59-111; the text fills
      max_seq_length, no padding), labels on the image slots, and the compressed attention mask straight from the ids."""
    B, T = text.shape
    dev = text.device
    key = (B, T, str(dev))
    if _GLUE.get("key") != key:
        _GLUE.update(key=key, offs=torch.arange(B + 1, device=dev, dtype=torch.int64) * T,
                     head=torch.tensor([IM_START, T2I], device=dev, dtype=torch.int64), tail=torch.zeros(0, device=dev, dtype=torch.int64))
    L = 2 + T + image_in.shape[1] + 2
    ids, _, labels = ops.t2i_assemble((text.reshape(-1), _GLUE["offs"]), image_in, image_labels, L, PAD, SOI, EOI, _GLUE["head"], _GLUE["tail"])
    return ids, labels, ops.mask_from_ids(ids, PAD, SOI, EOI, ops.MASK_T2I)


_GLUE = {}


def init_magvit_device(vq, seed):
    g = torch.Generator(device=next(vq.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, p in vq.named_parameters():
            if p.dim() == 4:
                p.normal_(0, 1.0 / math.sqrt(p.shape[1] * p.shape[2] * p.shape[3]), generator=g)
            elif "norm" in name and name.endswith("weight"):
                p.fill_(1.0)
            else:
                p.normal_(0, 0.05, generator=g)


def cpu_baseline(seq_len, sample_layers=4):
    """CPU oracle ("port") on a bounded sample: ONE t2i sample (L = seq_len) through `sample_layers`
    of the 28 1.5B-shape decoder layers (fwd+bwd, fp32 eager like BASELINE.md §2), the tied lm_head +
    CE on its 256 label rows, AdamW on those parameters, and MAGVITv2.get_code of one image; the
    layer/optimizer time is scaled by 28/sample_layers."""
    from oracle import magvit_ref, qwen2_ref, weights
    torch.manual_seed(0)
    cores = torch.get_num_threads()
    cfg = qwen2_ref.Qwen2Cfg(vocab_size=VOCAB, num_hidden_layers=sample_layers,
                             **{k: v for k, v in qwen2_ref.QWEN25_1P5B.items() if k != "num_hidden_layers"})
    lm = qwen2_ref.RefCausalLM(cfg)
    ids = torch.randint(0, TEXT_VOCAB, (1, seq_len))
    labels = torch.full((1, seq_len), -100)
    labels[:, -(NVQ + 1):-1] = torch.randint(TEXT_VOCAB, VOCAB - 1, (1, NVQ))
    opt = torch.optim.AdamW(lm.parameters(), lr=1e-4, weight_decay=0.01)
    times = {}

    def one():
        t0 = time.perf_counter()
        h = lm.backbone(ids, None, None)
        t1 = time.perf_counter()
        h_head = h.detach().requires_grad_(True)             # cut: head and stack backward are timed separately
        logits = lm.lm_head(h_head[:, -(NVQ + 1):-1])
        loss = torch.nn.functional.cross_entropy(logits.view(-1, VOCAB), labels[:, -(NVQ + 1):-1].reshape(-1))
        t2 = time.perf_counter()
        loss.backward()                                      # head: dlogits, tied-embedding wgrad, d(hidden)
        t3 = time.perf_counter()
        h.backward(h_head.grad)                              # stack: sample_layers decoder layers + embedding lookup
        t4 = time.perf_counter()
        opt.step()
        opt.zero_grad(set_to_none=True)
        t5 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4
    one()                                           # warm-up (allocator, Adam state)
    reps = sorted((one() for _ in range(3)), key=lambda t: sum(t))
    f, hd, b_head, b_stack, o = reps[1]              # the median of three repetitions
    n_layer = sum(p.numel() for p in lm.model.layers.parameters())
    n_head = lm.model.embed_tokens.weight.numel()
    scale = 28.0 / sample_layers
    layer_t = (f + b_stack) * scale                 # every term measured; only the layer count is scaled
    head_t = hd + b_head
    opt_t = o * (n_layer * scale + n_head) / (n_layer + n_head)     # AdamW is linear in the parameter count
    del lm, opt
    sd = weights.synth_magvit_state(magvit_ref.magvit_param_shapes(), seed=31)
    x = weights.synth_images(1, 256, seed=32)
    with torch.no_grad():
        t0 = time.perf_counter()
        magvit_ref.get_code_ref(sd, x)
        vq_t = time.perf_counter() - t0
    total = layer_t + head_t + opt_t + vq_t
    return {"value": round(1.0 / total, 5), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"median of 3 repetitions of 1 t2i sample L={seq_len} fp32 eager: {sample_layers}/28 decoder layers fwd+bwd (x{scale:.0f}), "
                      f"tied head+CE on 256 rows fwd+bwd (timed separately from the stack), AdamW (x param-count ratio), "
                      f"MAGVITv2.get_code(1 image); step {total:.1f}s = layers {layer_t:.1f} + head {head_t:.1f} + adamw {opt_t:.1f} + vq {vq_t:.1f}"}


def ar_decode_bench(model, dev, n_img=8, prefix=138, reps=2):
    """Second half of BASELINE.json's metric: AR image-tokens/s, BASELINE configs[3] shape (Best-of-N = 8 with CFG ->
    16 rows, prefix 138 = 128 text + template, 256 decode steps through the static-KV captured-graph path)."""
    model.eval()
    g = torch.Generator(device=dev).manual_seed(1)
    L = prefix + NVQ + 1
    ids = torch.randint(0, 151643, (n_img, L), device=dev, generator=g)
    un = torch.randint(0, 151643, (n_img, L), device=dev, generator=g)
    am = torch.ones((2 * n_img, L), dtype=torch.long, device=dev)
    best, times = None, []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = model.t2i_generate_ar(input_ids=ids, uncond_input_ids=un, attention_mask=am, guidance_scale=6.0,
                                     temperature=1.0, text_vocab_size=TEXT_VOCAB, image_token_num_per_image=NVQ)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        times.append(dt)
        best = dt if best is None else min(best, dt)
    model.train()
    bytes_step = 1310.3e6 * 2 + CODEBOOK * 1536 * 2                     # layer + code-head weights streamed once per step
    floor_ms = bytes_step / 6.3e12 * 1e3
    return {"value": round(n_img * NVQ / best, 1), "unit": "img-tokens/s", "images": n_img, "rows_with_cfg": 2 * n_img,
            "prefix": prefix, "decode_steps": NVQ, "hipgraph": bool(model.llm.engine.last_decode_graph),
            # a Best-of-N loop calls the generator once per prompt with the same shapes: call 1 captures the decode step, later calls
            # replay it (models/unigen.py: the session kept on the engine).  `value` is the steady state.
            "calls_tokens_per_s": [round(n_img * NVQ / t, 1) for t in times],
            "ms_per_step": round(best / NVQ * 1e3, 3),
            # weights streamed once per step against the 6.3 TB/s a pure stream achieves and against the 8 TB/s data-sheet peak
            "roofline": {"bound": "hbm", "floor_ms_per_step": round(floor_ms, 3), "frac": round(floor_ms / (best / NVQ * 1e3), 4),
                         "achieved": round(bytes_step / (best / NVQ) / 1e12, 3), "unit": "TB/s", "peak": 8.0,
                         "frac_of_8tb_peak": round(bytes_step / (best / NVQ) / 8e12, 4)}}


def extra_cases(model, vq, opt, dev, args, steps=5):
    """Secondary cases reported next to the headline (not `value`): the other BASELINE.json configs at their per-GPU shapes.
      * `t2i_L771_real_mask_ratio`: the headline step with the masking the reference applies (data/masking.py through the
        device kernel: t ~ U(0,1), cosine schedule, round(256 p) masked positions per sample) instead of mask_prob = 1;
      * `pt1_mixed_L387` (configs[1] as trained): configs/unigen_1_5b/unigen_pt1.yaml:87-93 -- 16 t2i + 8 mmu rows at L = 387;
      * `sft_L1603` (configs[2], per GPU): unigen_sft.yaml:69,96-98 -- 3 t2i + 1 lm + 4 mmu rows at L = 1603, SigLIP so400m
        tower on the 4 mmu images (384^2 -> 729 tokens), mm_projector, MAGVITv2 on the 3 t2i images;
      * `dpo_L387` (configs[4], per GPU): unigen_dpo.yaml:80, training/train_dpo.py:573-647 -- 10 chosen + 10 rejected image
        sequences, frozen reference-model forward, policy forward + get_batch_logps + backward + AdamW;
      * `maskgit_50_rounds` (configs[3]'s sibling sampler, scripts/run_evaluation.sh:195-197): UniGen.t2i_generate, 10 images with
        CFG 6 (20 rows), 50 rounds."""
    import types
    import torch.nn.functional as F
    from data.masking import mask_or_random_replace_tokens
    from models import UniGen
    from models.sampling import cosine_schedule
    from unigen_hip import ops
    from unigen_hip.dpo import get_batch_logps
    cfg = types.SimpleNamespace(training=types.SimpleNamespace(min_masking_rate=0.0, get=lambda k, d=None: d),
                                model=types.SimpleNamespace(codebook_size=CODEBOOK))
    g = torch.Generator(device=dev).manual_seed(SEED + 7)
    out = {}

    def timed(fn, n=steps):                 # five timed steps after two warm-ups (round 6: was two after one -- +-2 % by construction)
        fn()
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def by_family(fn):
        """one more pass of `fn` with HIP events around every library launch (on the launch stream), filed by kernel family:
        where a secondary case's time goes; ms per step"""
        from unigen_hip import lib as ug_lib
        torch.cuda.synchronize()
        ug_lib.PROFILE = {}
        fn()
        torch.cuda.synchronize()
        rec, ug_lib.PROFILE = ug_lib.PROFILE, None
        if os.environ.get("UNIGEN_BENCH_BY_LAUNCH"):           # the same events per entry point (GEMMs per shape), for tools/ and DESIGN
            per = {}
            for v in rec.values():
                for e0, e1, tag in v:
                    c = per.setdefault(tag, [0, 0.0])
                    c[0] += 1
                    c[1] += e0.elapsed_time(e1)
            os.makedirs("gpurun_out", exist_ok=True)
            with open(os.path.join("gpurun_out", "by_launch_%s.txt" % fn.__name__), "w") as f:
                for tag, (n, ms_) in sorted(per.items(), key=lambda kv: -kv[1][1]):
                    f.write("%9.3f ms %5d x %8.1f us  %s\n" % (ms_, n, ms_ / n * 1e3, tag))
        return {k: round(sum(e0.elapsed_time(e1) for e0, e1, _ in v), 2) for k, v in sorted(rec.items())}

    def cat_masks(*ms):
        return ops.MaskBits(torch.cat([m.bits for m in ms]), torch.cat([m.tileany for m in ms]), sum(m.B for m in ms), ms[0].L)

    # ---- headline shape, real mask ratios
    B = args.batch
    images = torch.rand(B, 3, 256, 256, device=dev, generator=g) * 2 - 1
    text = torch.randint(0, 151643, (B, args.text_len), device=dev, generator=g)
    n_lab = []

    def step_real():
        codes = vq.get_code(images) + TEXT_VOCAB
        ids_img, lab_img, _, _ = mask_or_random_replace_tokens(codes, MASK_ID, cfg, cosine_schedule)
        ids, labels, mask = t2i_rows(ops, text, ids_img, lab_img)
        _, l, _, _ = model(input_ids=ids, attention_mask=mask, labels=labels, batch_size_t2i=B, max_seq_length=args.text_len + 1,
                           num_vq_tokens=NVQ)
        l.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        n_lab.append((lab_img != -100).sum())
    ms = timed(step_real)
    out["t2i_L771_real_mask_ratio"] = {"ms_per_step": round(ms, 2), "samples_per_s": round(B / ms * 1e3, 2),
                                       "mean_label_rows": round(float(torch.stack(n_lab).float().mean()), 1), "of": B * NVQ}

    # ---- pt1 mixed batch
    Bt, Bm, T = 16, 8, 128
    L = T + NVQ + 3
    images = torch.rand(Bt + Bm, 3, 256, 256, device=dev, generator=g) * 2 - 1
    text_t = torch.randint(0, 151643, (Bt, T - 1), device=dev, generator=g)
    text_m = torch.randint(0, 151643, (Bm, L - NVQ - 4), device=dev, generator=g)
    MMU = 151670

    def step_pt1():
        codes = vq.get_code(images) + TEXT_VOCAB
        ids_img, lab_img, _, _ = mask_or_random_replace_tokens(codes[:Bt], MASK_ID, cfg, cosine_schedule)
        ids_t, lab_t, m_t = t2i_rows(ops, text_t, ids_img, lab_img)
        col = lambda v: torch.full((Bm, 1), v, device=dev)
        ids_m = torch.cat([col(IM_START), col(MMU), col(SOI), codes[Bt:], col(EOI), text_m], 1)       # mmu_prompt layout
        lab_m = ids_m.clone()
        lab_m[:, :NVQ + 4] = -100
        m_m = ops.mask_from_ids(ids_m, PAD, SOI, EOI, ops.MASK_MMU)                                     # create_attention_mask_for_mmu
        _, l_t, _, l_m = model(input_ids=torch.cat([ids_t, ids_m]), attention_mask=cat_masks(m_t, m_m), labels=torch.cat([lab_t, lab_m]),
                               batch_size_t2i=Bt, batch_size_mmu=Bm, max_seq_length=T, num_vq_tokens=NVQ)
        (l_t + l_m).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    ms = timed(step_pt1)
    # algorithmic FLOPs: t2i row at L = 387 3.475 TFLOP (+0.355 tokenizer); an mmu row carries the head on its
    # 386 shifted positions instead of 256 label rows: 3 x (1.0141 + 0.0186 + 386/256 x 0.1257) + 0.355
    fl = Bt * (3.475e12 + 0.3551e12) + Bm * (3 * (1.0141e12 + 0.0186e12 + 386 / 256 * 0.1257e12) + 0.3551e12)
    out["pt1_mixed_L387"] = {"ms_per_step": round(ms, 2), "samples_per_s": round((Bt + Bm) / ms * 1e3, 2), "rows": f"{Bt} t2i + {Bm} mmu",
                             "seq_len": L, "step_tflop": round(fl / 1e12, 2), "step_frac_of_bf16_peak": round(fl / (ms * 1e-3) / PEAK_BF16, 4),
                             "by_family_ms": by_family(step_pt1)}

    # ---- DPO: 10 pairs at L = 387 (a second, frozen copy of the model as the reference policy)
    pairs, beta = 10, 0.1
    Bd = 2 * pairs
    ref = UniGen(w_und_encoder=False, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                 codebook_size=CODEBOOK, num_vq_tokens=NVQ, load_from_pretrained=True, device=dev, init_seed=-1)
    ref.llm.init_weights_device(SEED)
    ref.eval().requires_grad_(False)
    images = torch.rand(Bd, 3, 256, 256, device=dev, generator=g) * 2 - 1                # chosen | rejected
    ids_d = torch.randint(0, 151643, (Bd, L), device=dev, generator=g)
    ids_d[pairs:, :L - NVQ - 2] = ids_d[:pairs, :L - NVQ - 2]                            # a pair shares its prompt
    ids_d[:, -(NVQ + 2)] = SOI
    ids_d[:, -1] = EOI
    dpo_loss = []

    def step_dpo():
        codes = vq.get_code(images) + TEXT_VOCAB
        msk = torch.rand(pairs, NVQ, device=dev, generator=g) < 0.6
        msk = torch.cat([msk, msk])                                                      # same masked positions within a pair
        ids_d[:, -(NVQ + 1):-1] = torch.where(msk, torch.full_like(codes, MASK_ID), codes)
        labels = torch.full((Bd, L), -100, device=dev)
        labels[:, -(NVQ + 1):-1] = torch.where(msk, codes, torch.full_like(codes, -100))
        mb = ops.mask_from_ids(ids_d, PAD, SOI, EOI, ops.MASK_T2I)
        with torch.no_grad():
            ref_lp = get_batch_logps(ref(input_ids=ids_d, attention_mask=mb, batch_size_t2i=Bd), labels, num_vq_tokens=NVQ)
        lp = get_batch_logps(model(input_ids=ids_d, attention_mask=mb, batch_size_t2i=Bd), labels, num_vq_tokens=NVQ)
        logits = (lp[:pairs] - lp[pairs:]) - (ref_lp[:pairs] - ref_lp[pairs:])
        loss = -F.logsigmoid(beta * logits).mean()
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        dpo_loss.append(loss.detach())
    ms = timed(step_dpo)
    # algorithmic FLOPs: every row is a t2i row at L = 387 (1.0141 linear + 0.0186 attention + 0.1257 head at its 256
    # label positions = 1.1584 TFLOP forward); the policy runs forward + backward (x 3), the frozen reference one more forward;
    # + the tokenizer's 0.3551 per image
    fl = Bd * ((3 + 1) * 1.1584e12 + 0.3551e12)
    out["dpo_L387"] = {"ms_per_step": round(ms, 2), "pairs_per_s": round(pairs / ms * 1e3, 2), "pairs": pairs, "seq_len": L,
                       "loss": round(float(dpo_loss[-1]), 4), "step_tflop": round(fl / 1e12, 2),
                       "step_frac_of_bf16_peak": round(fl / (ms * 1e-3) / PEAK_BF16, 4), "by_family_ms": by_family(step_dpo)}
    del ref
    torch.cuda.empty_cache()

    # ---- MaskGIT generation: 10 images, CFG 6, 50 rounds
    model.eval()
    n_img, prefix, rounds = 10, 138, 50
    Lg = prefix + NVQ + 1
    ids_g = torch.randint(0, 151643, (n_img, Lg), device=dev, generator=g)
    ids_g[:, prefix - 1] = SOI
    ids_g[:, -1] = EOI
    ids_g[:, prefix:-1] = MASK_ID
    un = ids_g.clone()
    un[:, :prefix - 10] = PAD
    mb = ops.mask_from_ids(torch.cat([ids_g, un]), PAD, SOI, EOI, ops.MASK_T2I)

    def gen():
        with torch.no_grad():
            return model.t2i_generate(input_ids=ids_g, uncond_input_ids=un, attention_mask=mb, guidance_scale=6.0, temperature=1.0,
                                      timesteps=rounds, generator=torch.Generator(device=dev).manual_seed(3),
                                      image_token_num_per_image=NVQ, text_vocab_size=TEXT_VOCAB)
    ms = timed(gen, n=1)
    out["maskgit_50_rounds"] = {"seconds_per_batch": round(ms / 1e3, 3), "images_per_s": round(n_img / ms * 1e3, 2), "images": n_img,
                                "rows_with_cfg": 2 * n_img, "rounds": rounds, "ms_per_round": round(ms / rounds, 2),
                                "img_tokens_per_s": round(n_img * NVQ / ms * 1e3, 1)}
    model.train()

    # ---- SFT mix at L = 1603 with the SigLIP tower (the projector joins the model and the optimizer here)
    from models.multimodal_encoder.siglip_encoder import SigLipVisionConfig, SigLipVisionTower
    Ls, bt, bl, bm, n_tok = 1603, 3, 1, 4, 729
    model.add_mm_projector(2, 1152)
    opt.add_param_group({"params": list(model.mm_projector.parameters()), "weight_decay": 0.01})
    tower = SigLipVisionTower("siglip-so400m-patch14-384", config=SigLipVisionConfig(patch_size=14)).to(dev).eval()
    with torch.no_grad():
        for p_ in tower.parameters():
            p_.normal_(0, 0.02, generator=g)
    images = torch.rand(bt, 3, 256, 256, device=dev, generator=g) * 2 - 1
    images_mmu = torch.rand(bm, 3, 384, 384, device=dev, generator=g) * 2 - 1
    ids_s = torch.randint(0, 151643, (bt + bl + bm, Ls), device=dev, generator=g)
    ids_s[:bt, -(NVQ + 2)] = SOI
    ids_s[:bt, -1] = EOI
    ids_s[:bt, -(NVQ + 1):-1] = MASK_ID
    embed = model.llm.model.embed_tokens
    r = torch.arange(Ls, device=dev)
    allow_lm = (r[None, :] <= r[:, None])[None].expand(bl, Ls, Ls)
    allow_mmu = ((r[None, :] <= r[:, None]) | ((r[None, :] >= 20) & (r[None, :] < 20 + n_tok)))[None].expand(bm, Ls, Ls)    # mmu_vit mask
    rest = ops.mask_compress(torch.cat([allow_lm, allow_mmu]))
    sft_loss = []

    def step_sft():
        codes = vq.get_code(images) + TEXT_VOCAB
        labels = torch.full((bt + bl + bm, Ls), -100, device=dev)
        labels[:bt, -(NVQ + 1):-1] = codes
        labels[bt:bt + bl] = ids_s[bt:bt + bl]
        labels[bt + bl:, 20 + n_tok:] = ids_s[bt + bl:, 20 + n_tok:]
        with torch.no_grad():
            feats = tower(images_mmu)                                       # [bm, 729, 1152] fp32
        img_h = model.mm_projector(feats)
        e = embed(ids_s)
        e = torch.cat([e[:bt + bl], torch.cat([e[bt + bl:, :20], img_h.float(), e[bt + bl:, 20 + n_tok:]], 1)])
        mb = cat_masks(ops.mask_from_ids(ids_s[:bt], PAD, SOI, EOI, ops.MASK_T2I), rest)
        _, l1, l2, l3 = model(input_ids=None, input_embeddings=e, attention_mask=mb, labels=labels, batch_size_t2i=bt,
                              batch_size_lm=bl, batch_size_mmu=bm, max_seq_length=Ls - NVQ - 3, num_vq_tokens=NVQ)
        (l1 + l2 + l3).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        sft_loss.append(torch.stack([l1.detach(), l2.detach(), l3.detach()]))
    ms = timed(step_sft)
    # algorithmic FLOPs (the mm_projector's 0.02 TFLOP per image is left out): linear 2.6204 GFLOP per token; attention
    # 172 032 FLOP per unmasked (q, k) pair over the 28 layers -- t2i row T (T + 1) / 2 + 258 (T + 258) with T = L - 258, lm row
    # L (L + 1) / 2, mmu_vit row the causal pairs + the image columns [20, 20 + 729) above the diagonal; head 0.4911 GFLOP per
    # position -- 256 label rows of a t2i row, the L - 1 shifted positions of an lm / mmu row (the convention of pt1_mixed_L387);
    # x 3 for forward + backward; + tokenizer 0.3551 per generation image and the frozen SigLIP tower 0.642 per understanding image
    T_ = Ls - 258
    pairs_t2i, pairs_lm = T_ * (T_ + 1) // 2 + 258 * (T_ + 258), Ls * (Ls + 1) // 2
    pairs_mmu = pairs_lm + 20 * n_tok + n_tok * (n_tok - 1) // 2
    fwd = ((bt + bl + bm) * Ls * 2.6204e9 + 172032.0 * (bt * pairs_t2i + bl * pairs_lm + bm * pairs_mmu)
           + 0.4911e9 * (bt * NVQ + (bl + bm) * (Ls - 1)))
    fl = 3 * fwd + bt * 0.3551e12 + bm * 0.642e12
    out["sft_L1603"] = {"ms_per_step": round(ms, 2), "samples_per_s": round((bt + bl + bm) / ms * 1e3, 2), "rows": f"{bt} t2i + {bl} lm + {bm} mmu",
                        "seq_len": Ls, "siglip_images": bm, "losses": [round(float(x), 3) for x in sft_loss[-1]],
                        "step_tflop": round(fl / 1e12, 2), "step_frac_of_bf16_peak": round(fl / (ms * 1e-3) / PEAK_BF16, 4),
                        "by_family_ms": by_family(step_sft)}
    model.llm.engine.check_errors()
    return out


RCCL_LOG = None            # NCCL_DEBUG_FILE of this rank (set before init_process_group at N > 1 on the nccl backend)


def rccl_debug_summary(path=None, text=None):
    """What RCCL said about the communicator at init (NCCL_DEBUG=INFO, subsystems INIT / GRAPH / TUNING, written to a per-rank
    file): channel counts, the ring / tree lines of this rank, the transports of its connections.  A single driver run at N > 1
    then tells ring from tree, how many channels the collectives use (each is a workgroup that takes a CU -- and watts -- from
    the GEMMs) and whether the links are xGMI (P2P/IPC) or something slower.  None when there is no log (gloo rehearsal)."""
    import re
    if text is None:
        path = path or RCCL_LOG
        if not path or not os.path.exists(path):
            return None
        text = open(path, errors="replace").read()
    lines = text.splitlines()
    pick = lambda pat: [re.sub(r"^.*?NCCL INFO ", "", l).strip() for l in lines if re.search(pat, l)]
    chan = pick(r"coll channels|nChannels|Channels? \d+ .*per")
    out = {"version": (pick(r"NCCL version|RCCL version") or [None])[0],
           "channels": chan[:4],
           "rings": pick(r"\bRing \d+ :")[:4], "trees": pick(r"\bTrees? \[")[:2],
           "connected": pick(r"Connected all (rings|trees)")[:4],
           "transports": sorted({m.group(1) for l in lines for m in [re.search(r"via (P2P/\S+|SHM\S*|NET/\S+|direct\S*)", l)] if m}),
           "algo_proto_table": pick(r"Algorithm|Algo/Proto|Latency/AlgBw")[:3],
           "log_lines": len(lines)}
    m = [int(x) for l in chan for x in re.findall(r"(\d+) coll channels", l)]
    out["coll_channels"] = m[0] if m else None
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks through torch.distributed.run as a
    child process and relay its stdout / exit code.  The parent must only ever SPAWN a child, never replace itself (no exec): on this
    pool an exec from a process that has touched the GPU runtime takes the machine down, and counting devices through HIP may
    initialise it -- so the devices are counted from sysfs (the KFD topology), without any HIP call."""
    import glob
    import socket
    import subprocess

    def gpu_nodes():
        n_gpu = 0
        for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            try:
                kv = dict(line.split(None, 1) for line in open(prop).read().splitlines() if " " in line)
                n_gpu += int(kv.get("simd_count", "0").strip()) > 0          # CPU nodes have no SIMDs
            except (OSError, ValueError):
                pass
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
        if vis:
            n_gpu = min(n_gpu, len([v for v in vis.split(",") if v.strip() != ""])) if n_gpu else len([v for v in vis.split(",") if v.strip() != ""])
        return n_gpu
    have = gpu_nodes()
    if have < n and os.environ.get("UNIGEN_BENCH_ONE_DEVICE") != "1":          # (no KFD topology in sysfs = no ROCm device at all)
        raise SystemExit(f"--gpus {n} but this node exposes {have} GPU(s); a weak-scaling number needs one GPU per rank "
                         "(UNIGEN_DIST_BACKEND=gloo UNIGEN_BENCH_ONE_DEVICE=1 rehearses the path on one device)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr is inherited
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


class PowerSampler:
    """Package power and shader clock of the busy GPU(s) while the timed steps run: a child process reads the hwmon files of every
    card every 20 ms (read-only sysfs; never touches the GPU); `report(t0, t1)` averages the cards that were busy in that window.
    profiles/r04_power.md: the step runs at the 1400 W cap with the clock pulled down -- the line says whether THIS box did too.
    Any failure (no hwmon, no permission) gives None; the benchmark never depends on it."""
    SRC = ("import sys, time, glob, os\n"
           "hw = [h for h in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')) if os.path.exists(h + '/power1_input')]\n"
           "f = open(sys.argv[1], 'w')\n"
           "while True:\n"
           "    row = []\n"
           "    for h in hw:\n"
           "        try:\n"
           "            row += [open(h + '/power1_input').read().strip(), open(h + '/freq1_input').read().strip()]\n"
           "        except (OSError, ValueError):\n"
           "            row += ['0', '0']\n"
           "    f.write('%.4f %s\\n' % (time.time(), ' '.join(row))); f.flush()\n"
           "    time.sleep(0.02)\n")

    def __init__(self):
        self.proc, self.path, self.cap = None, None, None
        try:
            import glob
            import subprocess
            import tempfile
            caps = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap")
            if not caps:
                return
            self.cap = int(open(caps[0]).read()) / 1e6
            self.path = os.path.join(tempfile.gettempdir(), f"unigen_bench_power_{os.getpid()}.txt")
            self.proc = subprocess.Popen([sys.executable, "-c", self.SRC, self.path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None

    def report(self, w0, w1):
        if self.proc is None:
            return None
        try:
            self.proc.terminate()
            self.proc.wait(timeout=5)
            cards = {}
            for line in open(self.path):
                try:                                             # (the last line may be cut off by the terminate above)
                    f = line.split()
                    if len(f) < 3 or len(f) % 2 == 0 or not (w0 <= float(f[0]) <= w1):
                        continue
                    row = [(int(f[1 + 2 * c]) / 1e6, int(f[2 + 2 * c]) / 1e6) for c in range((len(f) - 1) // 2)]
                except ValueError:
                    continue
                for c, smp_c in enumerate(row):
                    cards.setdefault(c, []).append(smp_c)
            os.remove(self.path)
            # (hwmon's power reading is itself a running average: the first third of the window still remembers what ran before)
            tails = [v[len(v) // 3:] for v in cards.values() if len(v) >= 3]
            means = [sum(s[0] for s in t) / len(t) for t in tails]
            if not means or max(means) < 300.0:
                return None
            busy = [t for t, m_ in zip(tails, means) if m_ >= 0.8 * max(means)]                      # the cards that ran the steps
            smp = [s for t in busy for s in t]
            return {"mean_w": round(sum(s[0] for s in smp) / len(smp), 1), "max_w": round(max(s[0] for s in smp), 1), "cap_w": self.cap,
                    "mean_sclk_mhz": round(sum(s[1] for s in smp) / len(smp), 1), "min_sclk_mhz": round(min(s[1] for s in smp), 1),
                    "nominal_sclk_mhz": 2400, "gpus_sampled": len(busy), "samples": len(smp),
                    "source": "hwmon power1_input / freq1_input every 20 ms over the last two thirds of the timed steps"}
        except Exception:
            return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="t2i samples per GPU (unigen_pt1.yaml: batch_size_t2i 16)")
    ap.add_argument("--text-len", type=int, default=511, help="text tokens after the 2 template tokens (513 total)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary cases (pt1 mixed batch, real mask ratios, SFT, DPO, MaskGIT)")
    ap.add_argument("--no-ar", action="store_true", help="skip the AR image-token generation measurement (second half of the metric)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    # UNIGEN_DIST_BACKEND=gloo + UNIGEN_BENCH_ONE_DEVICE=1: rehearse the N > 1 path with every rank on cuda:0 (RCCL refuses
    # two ranks on one device); the default is RCCL ("nccl") with one GPU per rank
    backend = os.environ.get("UNIGEN_DIST_BACKEND", "nccl")
    if os.environ.get("UNIGEN_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # RCCL's own account of the communicator (rings / trees / channels / transports) into a per-rank file that
            # rccl_debug_summary() condenses into the line's `exchange.rccl` -- unless the caller configured the debug output already
            global RCCL_LOG
            if "NCCL_DEBUG" not in os.environ and "NCCL_DEBUG_FILE" not in os.environ:
                RCCL_LOG = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"unigen_rccl_rank{rank}_{os.getpid()}.log")
                os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING", NCCL_DEBUG_FILE=RCCL_LOG)
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cpu = None
    L = 2 + args.text_len + NVQ + 2
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only (bench contract)
        cpu = cpu_baseline(L)

    from models import MAGVITv2, UniGen
    from unigen_hip import ops
    from unigen_hip.optim import FusedAdamW

    torch.manual_seed(SEED)
    model = UniGen(w_und_encoder=False, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                   codebook_size=CODEBOOK, num_vq_tokens=NVQ, load_from_pretrained=True, device=dev, init_seed=-1)
    model.llm.init_weights_device(SEED)              # same weights on every rank (DDP invariant)
    model.train()
    vq = MAGVITv2().to(dev).eval().requires_grad_(False)
    init_magvit_device(vq, SEED)
    decay = [p for n, p in model.named_parameters() if "bias" not in n]
    nodecay = [p for n, p in model.named_parameters() if "bias" in n]
    opt = FusedAdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}],
                     lr=1e-4, betas=(0.9, 0.999), eps=1e-8, overlap=True)     # update runs beside the next batch's tokenisation
    # N > 1: nothing to set up here -- the engine installs the flat-gradient exchange (unigen_hip/ddp.py) by itself on the
    # first backward of a process whose torch.distributed world is larger than one, and leaves the MEAN in the gradients

    B = args.batch
    g = torch.Generator(device=dev).manual_seed(SEED + rank)
    images = torch.rand(B, 3, 256, 256, device=dev, generator=g) * 2 - 1
    text = torch.randint(0, 151643, (B, args.text_len), device=dev, generator=g)
    losses = []

    def step():
        codes = vq.get_code(images) + TEXT_VOCAB
        masked = torch.full_like(codes, MASK_ID)                    # mask_prob = 1: every image token is a label
        ids, labels, mask = t2i_rows(ops, text, masked, codes)
        _, l_t2i, _, _ = model(input_ids=ids, attention_mask=mask, labels=labels, batch_size_t2i=B,
                               max_seq_length=args.text_len + 1, num_vq_tokens=NVQ)
        l_t2i.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(l_t2i.detach())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sampler = PowerSampler() if rank == 0 else None     # (started before the warm-up: it is sampling by the time the clock starts)
    for _ in range(args.warmup):
        step()
    model.llm.engine.check_errors()
    barrier()
    sync = model.llm.engine.grad_sync               # the flat-gradient exchange the engine installed (None at N = 1)
    if world > 1 and sync is None:
        raise SystemExit("world > 1 but the engine installed no gradient exchange: the timed steps were not data-parallel")

    def timed_block(nsteps):
        """nsteps steps between two barriers -> (max-over-ranks seconds, the exchange's own account of the block)"""
        wire0 = (sync.bytes_on_wire, sync.lookup_bytes_on_wire) if sync is not None else (0, 0)
        if sync is not None:
            sync.record_timeline = True             # event pairs around every bucket of every pass (no host syncs): read after the run
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        acct = None
        if world > 1:
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
            acct = dict(sync.describe(),
                        bytes_on_wire_per_step=int((sync.bytes_on_wire - wire0[0]) / nsteps),
                        lookup_bytes_on_wire_per_step=int((sync.lookup_bytes_on_wire - wire0[1]) / nsteps),
                        # the last timed step's buckets (rank 0): when each left, how long it queued, how long its collective took, and
                        # how long the compute stream stood waiting for the exchange at the end of backward (the exposed tail)
                        last_step=sync.timeline_report())
            sync.record_timeline = False
        return dt, acct

    w0 = time.time()
    dt, acct = timed_block(args.steps)
    power = sampler.report(w0, time.time()) if sampler is not None else None
    exchange = None
    if world > 1:
        # proof that the exchange's communicator spans N ranks: a SUM of ones on the process group the buckets move through
        exchange = dict(acct, ranks_seen=sync.ranks_seen(), early_embed_handovers=sync.early_embed_handovers, rccl=rccl_debug_summary())
        if exchange["ranks_seen"] != world:
            raise SystemExit(f"the exchange's communicator saw {exchange['ranks_seen']} ranks, expected {world}")
        # The driver's N > 1 run may be the only one this project gets: after the block that `value` is quoted on (the default exchange,
        # DDP's fp32 mean) every rank switches to bf16 on the wire with fp32 accumulation and times a second block of the same steps
        # (VERDICT r5 next 5).  Both accounts go under exchange.modes; `value` stays the first block's.
        modes = {sync.reduce: dict(samples_per_s=round(B * world / (dt / args.steps), 3), ms_per_step=round(dt / args.steps * 1e3, 2),
                                   exposed_wait_ms=(acct["last_step"] or {}).get("exposed_wait_ms"), last_step=acct["last_step"],
                                   bytes_on_wire_per_step=acct["bytes_on_wire_per_step"], collective=acct["collective"])}
        second = os.environ.get("UNIGEN_BENCH_SECOND_MODE", "bf16_fp32acc")
        if second and second != sync.reduce and sync.cuda:
            first = sync.reduce
            try:                                   # (the block `value` is quoted on is already measured: an error here must not cost the line)
                sync.set_reduce(second)
                for _ in range(2):
                    step()
                barrier()
                n2 = min(args.steps, 10)
                dt2, acct2 = timed_block(n2)
                modes[second] = dict(samples_per_s=round(B * world / (dt2 / n2), 3), ms_per_step=round(dt2 / n2 * 1e3, 2),
                                     exposed_wait_ms=(acct2["last_step"] or {}).get("exposed_wait_ms"), last_step=acct2["last_step"],
                                     bytes_on_wire_per_step=acct2["bytes_on_wire_per_step"], collective=acct2["collective"], steps=n2)
            except Exception as e:                 # noqa: BLE001 -- reported in the line, not raised
                modes[second] = {"error": f"{type(e).__name__}: {e}"[:400]}
            finally:
                sync.set_reduce(first)
        exchange["modes"] = modes
    ms = dt / args.steps * 1e3
    value = B * world / (dt / args.steps)

    # north_star's stated target is a fraction of the bf16 MFMA roofline "on the 1.5B fwd/bwd at 1 GPU": time the backbone's
    # forward + backward alone (same batch, tokens and mask prepared once; no tokenizer, no optimizer, no exchange), wall clock
    fwd_bwd = None
    if rank == 0 and world == 1 and not args.no_roofline:
        with torch.no_grad():
            codes = vq.get_code(images) + TEXT_VOCAB
            ids, labels, mask = t2i_rows(ops, text, torch.full_like(codes, MASK_ID), codes)
        opt.synchronize()

        def fb():
            _, l, _, _ = model(input_ids=ids, attention_mask=mask, labels=labels, batch_size_t2i=B,
                               max_seq_length=args.text_len + 1, num_vq_tokens=NVQ)
            l.backward()
            opt.zero_grad(set_to_none=True)
        fb()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fb()
        torch.cuda.synchronize()
        t_fb = (time.perf_counter() - t0) / 5
        fwd_bwd = {"ms": round(t_fb * 1e3, 2), "tflop": round(3 * B * FLOP_LLM_FWD / 1e12, 2),
                   "achieved": round(3 * B * FLOP_LLM_FWD / t_fb / 1e12, 1), "unit": "TFLOP/s",
                   "frac": round(3 * B * FLOP_LLM_FWD / t_fb / PEAK_BF16, 4),
                   "what": "UniGen forward + backward of the 28-layer backbone, head and loss on the bench batch (5 iterations, "
                           "wall clock); no tokenizer, optimizer or exchange"}

    roof = None
    if not args.no_roofline:
        # one more step with HIP events around every GEMM launch (on the launch stream): EVERY rank runs it -- the step
        # contains the gradient exchange -- and rank 0 records
        from unigen_hip import lib as ug_lib
        if rank == 0:
            ops.GEMM_PROFILE, ops.GEMM_PROFILE_FUSED = [], []
        step()
        torch.cuda.synchronize()
        # ... and one with events around EVERY library launch, filed by family (a separate step: nested inside the GEMM events
        # above, the second pair of event records per launch cost the GEMM figure 2-3 ms per step)
        rec_gemm, rec_fused = ops.GEMM_PROFILE, set(ops.GEMM_PROFILE_FUSED or [])
        ops.GEMM_PROFILE, ops.GEMM_PROFILE_FUSED = None, None
        if rank == 0:
            ug_lib.PROFILE = {}
        step()
        torch.cuda.synchronize()
        fam_rec, ug_lib.PROFILE = ug_lib.PROFILE, None
    if rank == 0 and not args.no_roofline:
        rec = rec_gemm
        tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in rec)
        tot_fl = sum(f for _, _, f in rec)
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM-side bytes per GEMM launch: PMC counters cannot be read from inside the benchmark process, so the value is
        # taken from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gemm_step_mix.py (the same GEMM mix) --
        # but ONLY when that record was made from this very gemm_bf16.hip (sha256 stored with it); a stale record is null
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic_current.json")
        if os.path.exists(tpath):
            import hashlib
            with open(tpath) as fh:
                rec_t = json.load(fh)
            with open(os.path.join(ROOT, "ml-unigen_amd", "csrc", "gemm_bf16.hip"), "rb") as fh:
                if rec_t.get("gemm_src_sha256") == hashlib.sha256(fh.read()).hexdigest():
                    traffic = rec_t.get("traffic_bytes_per_launch")
        roof = {"bound": "mfma", "kernel": "ug_gemm_bf16 family: gemm_kernel_p8 / _p8_wgrad_group / _p10 / gemm_kernel (bf16 MFMA GEMM)", "achieved": round(ach, 1), "peak": 2500.0,
                "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": traffic,
                "launches_per_step": len(rec), "avg_launch_ms": round(tot_ms / len(rec), 4),
                "flops_per_launch": round(tot_fl / len(rec) / 1e9, 2), "gemm_ms_per_step": round(tot_ms, 2),
                "step_frac_of_bf16_peak": round(value / world * FLOP_SAMPLE / PEAK_BF16, 4)}
        # since round 4 three launches per layer carry element-wise work in their epilogue (q/k/v + RoPE, gate_up + SwiGLU, down dgrad +
        # SwiGLU backward: ~5 ms per step that used to be separate HBM-bound kernels): `frac` above counts their whole duration
        # against their MFMA flops; the same ratio over the launches WITHOUT such an epilogue is reported next to it
        if rec_fused:
            p_ms = sum(e0.elapsed_time(e1) for i, (e0, e1, _) in enumerate(rec) if i not in rec_fused)
            p_fl = sum(f for i, (_, _, f) in enumerate(rec) if i not in rec_fused)
            f_ms = tot_ms - p_ms
            roof["plain_launches"] = {"launches": len(rec) - len(rec_fused), "ms_per_step": round(p_ms, 2), "achieved": round(p_fl / (p_ms * 1e-3) / 1e12, 1),
                                      "frac": round(p_fl / (p_ms * 1e-3) / 1e12 / 2500.0, 4)}
            roof["fused_epilogue_launches"] = {"launches": len(rec_fused), "ms_per_step": round(f_ms, 2), "achieved": round((tot_fl - p_fl) / (f_ms * 1e-3) / 1e12, 1),
                                               "frac": round((tot_fl - p_fl) / (f_ms * 1e-3) / 1e12 / 2500.0, 4),
                                               "what": "q/k/v projection + RoPE, gate_up + SwiGLU, down dgrad + SwiGLU backward: element-wise passes of rounds 1-3 (~11 ms per step) now inside these launches"}
        # launch time by kernel family in the same instrumented step (HIP events around every library launch, on the launch
        # stream) against each family's own bound; algorithmic work per step from SURVEY.md section 8d
        fam_ms = {k: sum(e0.elapsed_time(e1) for e0, e1, _ in v) for k, v in (fam_rec or {}).items()}
        n_par = model.llm.engine.fp.numel
        work = {"gemm": ("mfma", tot_fl, 2.5e15), "attention": ("mfma", 3 * B * 0.0569e12, 2.5e15),
                "tokenizer_and_towers": ("mfma f16 x 3 split products", B * 0.3551e12, 2.5e15 / 3),
                "adamw": ("hbm", 28.0 * n_par, 8e12), "elementwise": ("hbm", None, 8e12)}
        by_family = {}
        for k, t_ms in sorted(fam_ms.items(), key=lambda kv: -kv[1]):
            bound, w, peak = work[k]
            e = {"bound": bound, "ms_per_step": round(t_ms, 2), "launches": len(fam_rec[k])}
            if w is not None:
                e.update(achieved=round(w / (t_ms * 1e-3) / 1e12, 2), unit="TFLOP/s" if bound != "hbm" else "TB/s", frac=round(w / (t_ms * 1e-3) / peak, 4))
            by_family[k] = e
        roof["by_family"] = by_family
        roof["fwd_bwd_1p5b"] = fwd_bwd
    ar = None
    if rank == 0 and world == 1 and not args.no_ar:
        ar = ar_decode_bench(model, dev)
    extra = None
    if rank == 0 and world == 1 and not args.no_extra:
        extra = extra_cases(model, vq, opt, dev, args)           # (last: the SFT case adds the projector to the model)
    if rank == 0:
        out = {"metric": "train-step samples/s (1.5B, 256^2 img)", "value": round(value, 3), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "UniGen-1.5B stage-1 t2i pretrain step (BASELINE configs[1]): MAGVITv2 256^2->256 tokens + "
                                      f"{args.text_len + 2} text tokens, L={L}, per-GPU batch {B}, all 256 image tokens labelled, "
                                      "fwd+bwd+grad all-reduce+AdamW, random-init weights",
                          "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}"},
               "loss_first_last": [round(losses[0].item(), 4), round(losses[-1].item(), 4)],
               "ranks_seen": exchange["ranks_seen"] if exchange else 1, "exchange": exchange,
               "power": power, "roofline": roof, "cpu_baseline": cpu, "ar_decode": ar, "extra_cases": extra}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
