"""One DPO training step at full scale on ONE GPU (BASELINE configs[4] per-GPU shape: 10 chosen + 10 rejected image
sequences at L = 387 = 128 text + 256 image tokens + 3, reference training/train_dpo.py:573-647): MAGVITv2 codes of the 20
images, policy forward + fused get_batch_logps + backward + AdamW, frozen reference-model forward.  Random-init weights,
synthetic data.  A timing tool; the headline metric stays bench.py."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
import torch.nn.functional as F
from bench import CODEBOOK, EOI, MASK_ID, NVQ, PAD, SOI, TEXT_VOCAB, VOCAB, init_magvit_device
from models import MAGVITv2, UniGen
from unigen_hip import ops
from unigen_hip.dpo import get_batch_logps
from unigen_hip.optim import FusedAdamW


def main(steps=3):
    dev = torch.device("cuda:0")
    pairs, L, beta = 10, 128 + NVQ + 3, 0.1
    B = 2 * pairs

    def make(seed):
        m = UniGen(w_und_encoder=False, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                   codebook_size=CODEBOOK, num_vq_tokens=NVQ, device=dev, init_seed=-1)
        m.llm.init_weights_device(seed)
        return m

    policy = make(1).train()
    ref = make(1).eval().requires_grad_(False)
    vq = MAGVITv2().to(dev).eval().requires_grad_(False)
    init_magvit_device(vq, 1)
    opt = FusedAdamW([{"params": [p for p in policy.parameters() if p.requires_grad], "weight_decay": 0.01}], lr=1e-6)
    g = torch.Generator(device=dev).manual_seed(3)
    images = torch.rand(B, 3, 256, 256, device=dev, generator=g) * 2 - 1            # chosen | rejected
    ids = torch.randint(0, 151643, (B, L), device=dev, generator=g)
    ids[pairs:, :L - NVQ - 2] = ids[:pairs, :L - NVQ - 2]                           # a pair shares its prompt
    ids[:, -(NVQ + 2)] = SOI; ids[:, -1] = EOI
    times, losses = [], []
    for it in range(steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        codes = vq.get_code(images) + TEXT_VOCAB
        msk = torch.rand(pairs, NVQ, device=dev, generator=g) < 0.6
        msk = torch.cat([msk, msk])                                                # same masked positions within a pair
        ids[:, -(NVQ + 1):-1] = torch.where(msk, torch.full_like(codes, MASK_ID), codes)
        labels = torch.full((B, L), -100, device=dev)
        labels[:, -(NVQ + 1):-1] = torch.where(msk, codes, torch.full_like(codes, -100))
        mb = ops.mask_from_ids(ids, PAD, SOI, EOI, ops.MASK_T2I)
        with torch.no_grad():
            ref_lp = get_batch_logps(ref(input_ids=ids, attention_mask=mb, batch_size_t2i=B), labels, num_vq_tokens=NVQ)
        lp = get_batch_logps(policy(input_ids=ids, attention_mask=mb, batch_size_t2i=B), labels, num_vq_tokens=NVQ)
        logits = (lp[:pairs] - lp[pairs:]) - (ref_lp[:pairs] - ref_lp[pairs:])
        loss = -F.logsigmoid(beta * logits).mean()
        loss.backward()
        losses.append(round(loss.item(), 4))
        opt.step(); opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        policy.llm.engine.check_errors()
    print(json.dumps({"dpo_step_ms": round(min(times[1:]) * 1e3, 1), "first_step_ms": round(times[0] * 1e3, 1), "pairs": pairs, "L": L,
                      "losses": losses, "pairs_per_s": round(pairs / min(times[1:]), 2)}))


if __name__ == "__main__":
    main()
