"""AR image-token generation benchmark (BASELINE configs[3]: Best-of-N = 8 with CFG -> 16 rows, prefix 138,
256 decode steps, captured graph) on the full 1.5B-shape model with random-init weights."""
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


def run(n_img=8, prefix=138, n_tok=256, use_graph=True, reps=3):
    dev = torch.device("cuda:0")
    model = UniGen(w_und_encoder=False, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                   codebook_size=CODEBOOK, num_vq_tokens=n_tok, device=dev, init_seed=-1)
    model.llm.init_weights_device(10084)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(1)
    L = prefix + n_tok + 1
    ids = torch.randint(0, 151643, (n_img, L), device=dev, generator=g)
    un = torch.randint(0, 151643, (n_img, L), device=dev, generator=g)
    am = torch.ones((2 * n_img, L), dtype=torch.long, device=dev)
    best = None
    if os.environ.get("AR_PHASES"):                # where the call's wall time goes (adds host syncs: not the reported number)
        for _ in range(4):
            ph = {}
            model.t2i_generate_ar(input_ids=ids, uncond_input_ids=un, attention_mask=am, guidance_scale=6.0, temperature=1.0,
                                  text_vocab_size=TEXT_VOCAB, image_token_num_per_image=n_tok, use_graph=use_graph, timing=ph)
        print(json.dumps({"phases_ms": {k: round(v * 1e3, 3) for k, v in ph.items()}}), flush=True)
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = model.t2i_generate_ar(input_ids=ids, uncond_input_ids=un, attention_mask=am, guidance_scale=6.0, temperature=1.0,
                                     text_vocab_size=TEXT_VOCAB, image_token_num_per_image=n_tok, use_graph=use_graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert toks.shape == (n_img, n_tok) and int(toks.min()) >= 0 and int(toks.max()) < CODEBOOK
    return {"value": round(n_img * n_tok / best, 1), "unit": "img-tokens/s", "images": n_img, "rows_with_cfg": 2 * n_img,
            "prefix": prefix, "decode_steps": n_tok, "graph": use_graph, "seconds": round(best, 4),
            "ms_per_step": round(best / n_tok * 1e3, 3),
            "hbm_floor_ms_per_step": round((1310.3e6 * 2 + 8192 * 1536 * 2) / 6.3e12 * 1e3, 3)}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "graph":          # profiling runs: the captured path only, one repetition after the warm-up
        print(json.dumps(run(use_graph=True, reps=3)), flush=True)
    else:
        for ug in (False, True):
            print(json.dumps(run(use_graph=ug)), flush=True)
