"""MaskGIT (UniGen.t2i_generate) generation benchmark on the 1.5B-shape model, random-init weights: 8 images with CFG
(16 rows), prompt of 128 text tokens (+ template) -> L = 138 + 256 + 1, 18 rounds; incremental rounds vs full recompute."""
import json
import math
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from models import UniGen
from unigen_hip import ops

TEXT_VOCAB, CODEBOOK = 151674, 8192
VOCAB = TEXT_VOCAB + CODEBOOK + 1
PAD, SOI, EOI = 151665, 151666, 151667


def main(n_img=8, prefix=138, n=256, steps=18):
    dev = torch.device("cuda:0")
    model = UniGen(w_und_encoder=False, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                   codebook_size=CODEBOOK, num_vq_tokens=n, device=dev, init_seed=-1)
    model.llm.init_weights_device(10084)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(1)
    L = prefix + n + 1
    ids = torch.randint(0, 151643, (n_img, L), device=dev, generator=g)
    ids[:, prefix - 1] = SOI; ids[:, -1] = EOI; ids[:, prefix:-1] = VOCAB - 1
    un = ids.clone(); un[:, :prefix - 10] = PAD
    both = torch.cat([ids, un])
    # dense additive mask like create_attention_mask_predict_next(rm_pad_in_image=True) would build (here via the id kernel)
    mb = ops.mask_from_ids(both, PAD, SOI, EOI, ops.MASK_T2I)
    w = mb.bits.cpu()
    cols = torch.arange(mb.nW * 64)
    allow = ((w[:, :, cols // 64] >> (cols % 64)) & 1).bool()[:, :, :L]
    mask = torch.where(allow, 0.0, -1e30)[:, None].to(dev)
    res = {}
    for inc in (False, True):
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model.t2i_generate(input_ids=ids, uncond_input_ids=un, attention_mask=mask, guidance_scale=5.0, temperature=1.0,
                                     timesteps=steps, generator=torch.Generator(device=dev).manual_seed(3),
                                     image_token_num_per_image=n, text_vocab_size=TEXT_VOCAB, incremental=inc)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res["incremental" if inc else "full"] = {"seconds": round(best, 4), "images_per_s": round(n_img / best, 2),
                                                 "ms_per_round": round(best / steps * 1e3, 2)}
        res.setdefault("tokens", out.cpu())
        assert torch.equal(res["tokens"], out.cpu())
    del res["tokens"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
