"""AR decode: A/B of weight-prefetch plans (UNIGEN_DECODE_PREFETCH) on ONE model instance -- tokens/s per plan.  The plans only exist
with `tools/probes/decode_prefetch_r5.patch` applied (the experiment was measured slower and is not in the product:
profiles/r05_ar_prefetch.md); without the patch every plan is the plan-free run.  (Sampled tokens are NOT comparable between runs:
the split-K atomics make the logits' last bits run-dependent.)  Usage: python tools/ar_prefetch_sweep.py [plan ...]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from models import UniGen

TEXT_VOCAB, CODEBOOK = 151674, 8192
VOCAB = TEXT_VOCAB + CODEBOOK + 1
PLANS = sys.argv[1:] or ["off"]

dev = torch.device("cuda:0")
n_img, prefix, n_tok = 8, 138, 256
model = UniGen(w_und_encoder=False, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
               codebook_size=CODEBOOK, num_vq_tokens=n_tok, device=dev, init_seed=-1)
model.llm.init_weights_device(10084)
model.eval()
g = torch.Generator(device=dev).manual_seed(1)
L = prefix + n_tok + 1
ids = torch.randint(0, 151643, (n_img, L), device=dev, generator=g)
un = torch.randint(0, 151643, (n_img, L), device=dev, generator=g)
am = torch.ones((2 * n_img, L), dtype=torch.long, device=dev)


def run(plan, reps=3):
    os.environ["UNIGEN_DECODE_PREFETCH"] = plan
    best, toks, ph = None, None, {}
    for r in range(reps):
        gen = torch.Generator(device=dev).manual_seed(7)
        timing = {} if r == reps - 1 else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = model.t2i_generate_ar(input_ids=ids, uncond_input_ids=un, attention_mask=am, guidance_scale=6.0, temperature=1.0,
                                     text_vocab_size=TEXT_VOCAB, image_token_num_per_image=n_tok, generator=gen,
                                     **({"timing": timing} if timing is not None else {}))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if timing is None:
            best = dt if best is None else min(best, dt)
        else:
            ph = timing
    return best, toks, ph


base_t, base_toks, _ = run("off")
for plan in PLANS:
    t, toks, ph = run(plan)
    print(json.dumps({"plan": plan, "tok_per_s": round(n_img * n_tok / t, 1), "ms_per_step": round(t / n_tok * 1e3, 4),
                      "replay_ms_per_step": round(ph.get("replay", 0.0) / (n_tok - 2) * 1e3, 4),
                      "tokens_equal_plan_free": bool(torch.equal(toks, base_toks))}), flush=True)
