"""Understanding-side generation at full scale on ONE GPU: R prompts of 768 tokens (729 image-feature positions + text,
the CoT-V rating shape, reference evaluation/inference_unigen_cot.py:308-415), 16 new tokens each, greedy: R calls of
mmu_generate (KV-cached) against one mmu_generate_batch.  Random-init weights, synthetic prompts."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from bench import CODEBOOK, NVQ, TEXT_VOCAB, VOCAB
from models import UniGen


def main():
    dev = torch.device("cuda:0")
    R, L, new = int(os.environ.get("R", "16")), 768, 16
    model = UniGen(w_und_encoder=False, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                   codebook_size=CODEBOOK, num_vq_tokens=NVQ, device=dev, init_seed=-1)
    model.llm.init_weights_device(1)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(2)
    idx = torch.randint(0, 151643, (R, L), device=dev, generator=g)
    r = torch.arange(L, device=dev)
    allow = (r[None, :] <= r[:, None]) | ((r[None, :] >= 20) & (r[None, :] < 749))      # mmu_vit mask: image block visible
    mask = torch.where(allow, 0.0, torch.finfo(torch.float32).min)[None, None].expand(R, 1, L, L).contiguous()

    def timed(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    t_seq, seq = timed(lambda: [model.mmu_generate(idx=idx[i:i + 1], attention_mask=mask[i:i + 1], max_new_tokens=new,
                                                   temperature=0.0) for i in range(R)])
    t_bat, bat = timed(lambda: model.mmu_generate_batch(idx=idx, attention_mask=mask, max_new_tokens=new, temperature=0.0))
    same = sum(int([int(t) for t in a][:4] == [int(t) for t in b][:4]) for a, b in zip(seq, bat))
    print(json.dumps({"rows": R, "prompt": L, "new_tokens": new, "sequential_s": round(t_seq, 3), "batched_s": round(t_bat, 3),
                      "speedup": round(t_seq / t_bat, 2), "rows_with_equal_first_4_tokens": same}))


if __name__ == "__main__":
    main()
