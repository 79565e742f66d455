"""Child process of tests/test_ddp_gpu.py::test_rccl_two_gpus_every_transport_and_wire_format (needs >= 2 GPUs; never imported
by pytest).

    python ddp_rccl_worker.py <rank> <world> <port> <out.pt>

One rank per GPU over RCCL ("nccl").  Drives unigen_hip.ddp.FlatGradSync on a synthetic flat gradient buffer with the hook order
backward produces -- every transport (torch.distributed / the library's ug_comm_* entry points) x every wire format -- including
the tied table's early dense hand-over and the lookups' (id, row) exchange, and records what each rank holds afterwards."""
import os
import sys
import types

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-unigen_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
from test_ddp_cpu import _synthetic_rank_grads  # noqa: E402
from unigen_hip.ddp import FlatGradSync  # noqa: E402

n_layers, per_layer, V, H, norm = 4, 100003 // 64 * 64 + 64, 500, 64, 64
embed = V * H
off = {"embed": (0, (V, H))}
pos = embed
for i in range(n_layers):
    off[f"l{i}.wqkv"] = (pos, (per_layer,))
    pos += per_layer
off["norm"] = (pos, (norm,))
numel = pos + norm
local = _synthetic_rank_grads(world, n=numel, seed=9)[rank].to(dev)
gen = torch.Generator().manual_seed(50 + rank)
n_rows = 37 + 5 * rank                                     # ranks look up different numbers of rows
ids = torch.randint(0, V, (n_rows,), generator=gen)
ids[: n_rows // 3] = 7
rows = torch.randn(n_rows, H, generator=gen)
res = {"local": local.cpu(), "ids": ids, "rows": rows, "cases": {}}
for transport in ("torch", "ug_comm"):
    for reduce in ("fp32", "fp32_rsag", "bf16_fp32acc", "bf16"):
        os.environ["UNIGEN_DDP_TRANSPORT"] = transport
        grad = local.clone()
        eng = types.SimpleNamespace(fp=types.SimpleNamespace(grad=grad, off=off), dims=types.SimpleNamespace(num_hidden_layers=n_layers),
                                    grad_ready_hook=None)
        sync = FlatGradSync(eng, layers_per_bucket=2, reduce=reduce)
        seen = sync.ranks_seen()
        sync.begin(lookup_rows=n_rows)
        eng.grad_ready_hook("head")
        eng.grad_ready_hook("norm")
        for i in reversed(range(n_layers)):
            eng.grad_ready_hook(i)
        sync.add_lookup(ids.to(dev), rows.to(dev))
        eng.grad_ready_hook("embed")
        sync.finish()
        torch.cuda.synchronize()
        res["cases"][f"{transport}:{reduce}"] = dict(grad=grad.cpu(), seen=seen, bytes=sync.bytes_on_wire, lookup_bytes=sync.lookup_bytes_on_wire,
                                                     early=sync.early_embed_handovers, backend=sync.backend)
        del sync
torch.save(res, out)
dist.barrier()
dist.destroy_process_group()
