"""One SFT-mix training step at full scale on ONE GPU (BASELINE configs[2] per-GPU shape: 3 t2i + 1 lm + 4 mmu rows at
L = 1603, SigLIP so400m tower on 4 images of 384^2, mm_projector, MAGVITv2 on 3 images), random-init weights, synthetic
data.  A sanity / timing tool for the understanding branch; the headline metric stays bench.py."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from bench import CODEBOOK, EOI, MASK_ID, NVQ, PAD, SOI, TEXT_VOCAB, VOCAB, init_magvit_device
from models import MAGVITv2, UniGen
from models.multimodal_encoder.siglip_encoder import SigLipVisionConfig, SigLipVisionTower
from unigen_hip import ops
from unigen_hip.optim import FusedAdamW


def main(steps=3):
    dev = torch.device("cuda:0")
    L, bt, bl, bm, n_img = 1603, 3, 1, 4, 729
    model = UniGen(w_und_encoder=True, vocab_size=VOCAB, llm_vocab_size=TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                   codebook_size=CODEBOOK, num_vq_tokens=NVQ, mm_input_dim=1152, und_proj_depth=2, device=dev, init_seed=-1)
    model.llm.init_weights_device(1)
    model.mm_projector.to(dev)
    model.train()
    vq = MAGVITv2().to(dev).eval().requires_grad_(False)
    init_magvit_device(vq, 1)
    tower = SigLipVisionTower("siglip-so400m-patch14-384", config=SigLipVisionConfig(patch_size=14)).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for p in tower.parameters():
            p.normal_(0, 0.02, generator=g)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdamW([{"params": params, "weight_decay": 0.01}], lr=1e-5)
    images = torch.rand(bt, 3, 256, 256, device=dev, generator=g) * 2 - 1
    images_mmu = torch.rand(bm, 3, 384, 384, device=dev, generator=g) * 2 - 1
    ids = torch.randint(0, 151643, (bt + bl + bm, L), device=dev, generator=g)
    ids[:bt, -(NVQ + 2)] = SOI; ids[:bt, -1] = EOI
    embed = model.llm.model.embed_tokens
    times = []
    for it in range(steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        codes = vq.get_code(images) + TEXT_VOCAB
        ids[:bt, -(NVQ + 1):-1] = MASK_ID
        labels = torch.full((bt + bl + bm, L), -100, device=dev)
        labels[:bt, -(NVQ + 1):-1] = codes
        labels[bt:bt + bl] = ids[bt:bt + bl]
        labels[bt + bl:, 20 + n_img:] = ids[bt + bl:, 20 + n_img:]
        with torch.no_grad():
            feats = tower(images_mmu)                                   # [bm, 729, 1152] fp32
        img_h = model.mm_projector(feats)
        e = embed(ids)
        e = torch.cat([e[:bt + bl], torch.cat([e[bt + bl:, :20], img_h.float(), e[bt + bl:, 20 + n_img:]], 1)])
        m_t2i = ops.mask_from_ids(ids[:bt], PAD, SOI, EOI, ops.MASK_T2I)
        # one MaskBits for the mixed batch: t2i rows | causal lm row | mmu_vit rows (prefix 20, image columns visible)
        r = torch.arange(L, device=dev)
        allow_lm = (r[None, :] <= r[:, None])[None].expand(bl, L, L)
        allow_mmu = ((r[None, :] <= r[:, None]) | ((r[None, :] >= 20) & (r[None, :] < 20 + n_img)))[None].expand(bm, L, L)
        rest = ops.mask_compress(torch.cat([allow_lm, allow_mmu]))
        mb = ops.MaskBits(torch.cat([m_t2i.bits, rest.bits]), torch.cat([m_t2i.tileany, rest.tileany]), bt + bl + bm, L)
        _, l1, l2, l3 = model(input_ids=None, input_embeddings=e, attention_mask=mb, labels=labels, batch_size_t2i=bt,
                              batch_size_lm=bl, batch_size_mmu=bm, max_seq_length=L - NVQ - 3, num_vq_tokens=NVQ)
        (l1 + l2 + l3).backward()
        opt.step(); opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        model.llm.engine.check_errors()
    print(json.dumps({"sft_step_ms": round(min(times[1:]) * 1e3, 1), "first_step_ms": round(times[0] * 1e3, 1), "rows": bt + bl + bm,
                      "L": L, "losses": [round(float(x), 3) for x in (l1, l2, l3)], "samples_per_s": round((bt + bl + bm) / min(times[1:]), 2)}))


if __name__ == "__main__":
    main()
