"""Which host call sites launch torch fill / copy kernels inside one bench step (torch.profiler with stacks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-unigen_amd")]
import torch
import bench
from models import MAGVITv2, UniGen
from unigen_hip import ops
from unigen_hip.optim import FusedAdamW
dev = torch.device("cuda:0")
model = UniGen(w_und_encoder=False, vocab_size=bench.VOCAB, llm_vocab_size=bench.TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
               codebook_size=bench.CODEBOOK, num_vq_tokens=bench.NVQ, load_from_pretrained=True, device=dev, init_seed=-1)
model.llm.init_weights_device(1)
model.train()
vq = MAGVITv2().to(dev).eval().requires_grad_(False)
bench.init_magvit_device(vq, 1)
opt = FusedAdamW(model.parameters(), lr=1e-4, overlap=True)
B = 16
images = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
text = torch.randint(0, 151643, (B, 511), device=dev)


def step():
    codes = vq.get_code(images) + bench.TEXT_VOCAB
    ids, labels, mask = bench.t2i_rows(ops, text, torch.full_like(codes, bench.MASK_ID), codes)
    _, l, _, _ = model(input_ids=ids, attention_mask=mask, labels=labels, batch_size_t2i=B, max_seq_length=512, num_vq_tokens=bench.NVQ)
    l.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
import collections
agg = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::full", "aten::empty_like", "aten::clone", "aten::to", "aten::_to_copy", "aten::cat",
                   "aten::add", "aten::mul", "aten::index", "aten::arange"):
        st = [s for s in (ev.stack or []) if "ml-unigen_amd" in s or "bench.py" in s]
        key = (ev.name, st[0].split("ml-unigen_amd/")[-1] if st else "?")
        agg[key] += 1
        dur[key] += ev.device_time_total
for k, c in agg.most_common(40):
    print(f"{c:4d} x {k[0]:18s} device {dur[k]:9.1f} us  {k[1][:120]}")
