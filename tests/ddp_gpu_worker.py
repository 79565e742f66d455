"""Child process of tests/test_ddp_gpu.py (started fresh with subprocess; never imported by pytest).

    python ddp_gpu_worker.py <rank> <world> <port> <mode> <out.pt>

Every rank drives cuda:0 (RCCL refuses two ranks on one device, so the process group is gloo: the collective moves through
host memory but the whole product path -- autograd hooks, buckets, side stream, pack / unpack kernels, end-of-backward
callback, optimizer -- is the one an 8-GPU run executes).  mode:
  single  world 1, no process group: the reference result on the concatenated batch
  ddp     the model wrapped in torch DistributedDataParallel exactly as accelerator.prepare does (train.py:492), stock
          torch.optim.AdamW, clip_grad_norm_ between backward and step (train.py:775-780)
  bare    no wrapper (bench.py's situation), FusedAdamW, plus a gradient-accumulation step under UniGen.no_sync()
  multi   no wrapper; every forward looks embeddings up TWICE (two parts concatenated, as the reference's callers do,
          training/train.py:602-609,633,671) and every backward covers TWO decoder-stack segments (the local batch as two
          micro-batches under one backward) -- ADVICE r2 (high): the embedding bucket must be exchanged once, after its last
          writer, and layer buckets only by the last segment
"""
import os
import sys

rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-unigen_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from helpers import additive, golden, llm_config_dir  # noqa: E402
from models import UniGen  # noqa: E402
from oracle import host_ref, weights  # noqa: E402   (test-side: seeded weights and the mask builder)

dev = torch.device("cuda:0")
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
g = golden("g2_tiny_unigen.pt")
cfg, ids = g["cfg"], g["ids"]
W_UND = mode == "bare"                                   # the bare case also carries ordinary (non-flat) parameters
model = UniGen(w_und_encoder=W_UND, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=llm_config_dir(cfg),
               codebook_size=20, num_vq_tokens=16, load_from_pretrained=True, device=dev, init_seed=1, mm_input_dim=32, und_proj_depth=2)
names = [(n, tuple(p.shape)) for n, p in model.named_parameters() if n != "_ddp_anchor"]
sd = weights.synth_llm_state(names, seed=g["weight_seed"] + (0 if rank == 0 else 17))      # rank 1 starts DIFFERENT: the
model.load_state_dict(sd, strict=False)                                                    # install-time broadcast must fix it
model.train()

Bh, L, n, STEPS = 2, 40, 16, 3
full = Bh * max(world, 2)


def batch(step):
    gen = torch.Generator().manual_seed(1000 + step)
    seq = torch.randint(0, 290, (full, L), generator=gen)
    seq[:, -(n + 2)] = ids["soi"]; seq[:, -1] = ids["eoi"]
    seq[1, :5] = ids["pad"]; seq[2, :9] = ids["pad"]
    img = torch.randint(312, 332, (full, n), generator=gen)
    labels = torch.full((full, L), -100)
    labels[:, -(n + 1):-1] = img                           # every image position labelled: equal label counts per rank,
    seq[:, -(n + 1):-1] = ids["mask"]                      # so the mean of the per-rank means is the global mean
    mask = additive(host_ref.mask_predict_next_ref(seq, ids["pad"], ids["soi"], ids["eoi"], rm_pad_in_image=True))
    feats = torch.randn(full, 6, 32, generator=gen)
    sl = slice(0, full) if world == 1 else slice(rank * Bh, (rank + 1) * Bh)
    return seq[sl].to(dev), mask[sl].to(dev), labels[sl].to(dev), feats[sl].to(dev)


def groups(named):
    named = [(k, p) for k, p in named]
    return [{"params": [p for k, p in named if "bias" not in k], "weight_decay": 0.01},
            {"params": [p for k, p in named if "bias" in k], "weight_decay": 0.0}]


net = model
if mode == "ddp":
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
    assert model._is_ddp_wrapped()
    opt = torch.optim.AdamW(groups(net.named_parameters()), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
else:
    from unigen_hip.optim import FusedAdamW
    opt = (FusedAdamW if mode == "bare" else torch.optim.AdamW)(groups(model.named_parameters()), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)

rec = {"losses": [], "mode": mode, "rank": rank}
probe = ["llm.model.layers.0.self_attn.q_proj.weight", "llm.model.layers.1.mlp.down_proj.weight", "llm.model.norm.weight",
         "llm.model.embed_tokens.weight", "llm.model.layers.1.self_attn.k_proj.bias"]
params = dict(model.named_parameters())


def loss_of(seq, mask, labels, feats):
    if mode == "multi":
        total, h = 0.0, seq.shape[0] // 2
        for sl in (slice(0, h), slice(h, seq.shape[0])):
            emb = torch.cat([model.llm.model.embed_tokens(seq[sl, :17]), model.llm.model.embed_tokens(seq[sl, 17:])], 1)
            _, l1, _, _ = net(input_ids=seq[sl], input_embeddings=emb, attention_mask=mask[sl], labels=labels[sl], batch_size_t2i=h,
                              max_seq_length=L - n - 3, num_vq_tokens=n)
            total = total + 0.5 * l1
        return total
    if W_UND:                                              # projector output replaces 6 text embeddings: ordinary parameters
        emb = model.llm.model.embed_tokens(seq)            # on the graph next to the flat ones
        emb = torch.cat([emb[:, :10], model.mm_projector(feats).to(emb.dtype), emb[:, 16:]], 1)
        _, l1, _, _ = net(input_ids=seq, input_embeddings=emb, attention_mask=mask, labels=labels, batch_size_t2i=seq.shape[0],
                          max_seq_length=L - n - 3, num_vq_tokens=n)
    else:
        _, l1, _, _ = net(input_ids=seq, attention_mask=mask, labels=labels, batch_size_t2i=seq.shape[0], max_seq_length=L - n - 3,
                          num_vq_tokens=n)
    return l1


for step in range(STEPS):
    l1 = loss_of(*batch(step))
    opt.zero_grad(set_to_none=True)
    l1.backward()
    if step == 0:
        rec["grads0"] = {k: params[k].grad.detach().float().cpu().clone() for k in probe}
        if W_UND:
            rec["grads0"]["mm_projector.0.weight"] = params["mm_projector.0.weight"].grad.detach().float().cpu().clone()
    rec["gnorm%d" % step] = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0))
    opt.step()
    rec["losses"].append(float(l1))

if mode == "bare" and world > 1:
    # gradient accumulation: a micro-step under no_sync() exchanges nothing, the next backward exchanges the SUM of both
    sync = model.llm.engine.grad_sync
    opt.zero_grad(set_to_none=True)
    before = sync.bytes_on_wire
    with model.no_sync():
        loss_of(*batch(10)).backward()
    rec["no_sync_bytes"] = sync.bytes_on_wire - before
    local = params[probe[0]].grad.detach().float().cpu().clone()
    loss_of(*batch(11)).backward()
    rec["accum_bytes"] = sync.bytes_on_wire - before
    rec["accum_grad"] = params[probe[0]].grad.detach().float().cpu().clone()
    rec["accum_local_first"] = local

rec["weights"] = {k: params[k].detach().float().cpu().clone() for k in probe}
eng = model.llm.engine
rec["sync"] = None if eng.grad_sync is None else dict(reduce=eng.grad_sync.reduce, bytes=eng.grad_sync.bytes_on_wire,
                                                      numel=eng.fp.grad.numel(), backend=eng.grad_sync.backend,
                                                      lookup_bytes=eng.grad_sync.lookup_bytes_on_wire,
                                                      early=eng.grad_sync.early_embed_handovers)
torch.cuda.synchronize()
torch.save(rec, out)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
