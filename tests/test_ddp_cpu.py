"""N > 1 path on CPU: two gloo processes drive unigen_hip.ddp.FlatGradSync over a flat gradient buffer
with the backbone's real layout (tiny dims) and the hook order backward produces; after finish() every
rank must hold the MEAN of the per-rank gradients (what DDP leaves in .grad), also for ordinary parameters handed
in as `extra_params`; a pass begun with enabled=False (no_sync) must exchange nothing."""
import os
import socket
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, layers_per_bucket, q):
    sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unigen_hip.ddp import FlatGradSync
    n_layers, per_layer, embed, norm = 6, 1000, 3000, 64
    off = {"embed": (0, (embed,))}
    pos = embed
    for i in range(n_layers):
        off[f"l{i}.wqkv"] = (pos, (per_layer,))
        pos += per_layer
    off["norm"] = (pos, (norm,))
    numel = pos + norm
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(numel, generator=g)
    mine = grad.clone()
    eng = types.SimpleNamespace(fp=types.SimpleNamespace(grad=grad, off=off), dims=types.SimpleNamespace(num_hidden_layers=n_layers),
                                grad_ready_hook=None)
    extra = torch.nn.Parameter(torch.zeros(7))
    extra.grad = torch.full((7,), float(rank + 1))
    sync = FlatGradSync(eng, layers_per_bucket=layers_per_bucket, extra_params=lambda: [extra])
    assert sync.reduce == "fp32"
    # backward order: final norm, layers N-1..0, embedding
    sync.begin()
    eng.grad_ready_hook("norm")
    for i in reversed(range(n_layers)):
        eng.grad_ready_hook(i)
    eng.grad_ready_hook("embed")
    sync.finish()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = torch.stack(gathered).mean(0)
    ok = torch.allclose(grad, want, atol=1e-6) and sync.grad_scale == 1.0
    ok = ok and torch.allclose(extra.grad, torch.full((7,), (1 + world) / 2.0)) and sync.bytes_on_wire == (numel + 7) * 4
    # second step, inputs_embeds path: the embedding hook never fires, finish() must flush the head itself
    grad.copy_(mine)
    sync.begin()
    eng.grad_ready_hook("norm")
    for i in reversed(range(n_layers)):
        eng.grad_ready_hook(i)
    sync.finish()
    ok = ok and torch.allclose(grad, want, atol=1e-6)
    # third pass, gradient-accumulation micro-step: nothing moves, local gradients stay
    grad.copy_(mine)
    sent = sync.bytes_on_wire
    sync.begin(enabled=False)
    eng.grad_ready_hook("norm")
    for i in reversed(range(n_layers)):
        eng.grad_ready_hook(i)
    eng.grad_ready_hook("embed")
    sync.finish()
    ok = ok and torch.equal(grad, mine) and sync.bytes_on_wire == sent
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("layers_per_bucket", [1, 4])
def test_flat_grad_sync_two_ranks(layers_per_bucket):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, layers_per_bucket, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
