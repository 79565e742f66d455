"""N > 1 rehearsal on ONE GPU: two fresh child processes share cuda:0 under a gloo process group and run
3 optimizer steps of the tiny UniGen through the real product path -- auto-installed FlatGradSync (hooks -> buckets -> side
stream -> pack / all-reduce / unpack -> end-of-backward callback) -- once wrapped in torch DistributedDataParallel with a stock
torch.optim.AdamW and clip_grad_norm_ exactly like training/train.py:492,775-780, once bare with FusedAdamW (bench.py's
situation).  Gate: the averaged gradients equal the single-process gradients on the concatenated
batch within bf16 reduction error (1e-2 relative; 2e-4 with the fp32 exchange), and the loss curves agree within 1e-3."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _run(world, mode, tmp_path, reduce):
    port = _free_port()
    env = dict(os.environ, UNIGEN_DDP_REDUCE=reduce, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / f"{mode}_{reduce}_{r}.pt") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_gpu_worker.py"), str(r), str(world), port, mode, outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(o, weights_only=False) for o in outs]


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("reduce", ["fp32", "fp32_rsag", "bf16_fp32acc", "bf16"])
def test_two_ranks_on_one_gpu_match_single_process(dev, tmp_path, reduce):
    tol = 2e-4 if reduce.startswith("fp32") else 1e-2
    for mode in ("ddp", "bare", "multi"):
        single = _run(1, "single" if mode == "ddp" else mode, tmp_path, reduce)[0]
        r0, r1 = _run(2, mode, tmp_path, reduce)
        assert r0["sync"]["reduce"] == reduce and r0["sync"]["backend"] == "gloo"
        worst = 0.0
        for k, want in single["grads0"].items():
            for r in (r0, r1):
                worst = max(worst, _rel(r["grads0"][k], want))
        # both ranks hold the same averaged gradient and, three steps later, the same weights
        for k in r0["grads0"]:
            assert torch.equal(r0["grads0"][k], r1["grads0"][k]), (mode, k)
        for k in r0["weights"]:
            assert torch.equal(r0["weights"][k], r1["weights"][k]), (mode, k)
        mean_loss = [(a + b) / 2 for a, b in zip(r0["losses"], r1["losses"])]
        lerr = max(abs(a - b) / b for a, b in zip(mean_loss, single["losses"]))
        gn = abs(r0["gnorm0"] - single["gnorm0"]) / single["gnorm0"]
        print(f"[{mode}, {reduce}] grads vs single process: worst rel {worst:.2e}; loss curve rel {lerr:.2e}; grad-norm rel {gn:.2e}; "
              f"{r0['sync']['bytes'] / 1e6:.1f} MB exchanged")
        assert worst < tol, (mode, reduce, worst)
        assert lerr < 1e-3 and gn < tol
        per_step = r0["sync"]["numel"] * (2 if reduce == "bf16" else 4)
        # the tied table's dense part goes early, the lookups as (id, row) pairs: 2 ranks x cap rows x (8-byte id + fp32 row)
        assert r0["sync"]["lookup_bytes"] > 0 and r0["sync"]["lookup_bytes"] % (2 * (8 + 4 * 256)) == 0
        if reduce == "bf16_fp32acc":                 # (bucket slices are padded to 8 elements per rank: not an exact multiple)
            assert r0["sync"]["bytes"] >= 3 * per_step
            continue
        if mode == "bare":
            # 3 steps + 1 exchanged accumulation step (+ the ordinary parameters, a few KB); the no_sync step moved nothing
            assert r0["no_sync_bytes"] == 0 and r0["accum_bytes"] > 0
            extras = 4 * 4 * (32 * 256 + 256 + 256 * 256 + 256)          # mm_projector gradients, fp32, 4 exchanges
            assert r0["sync"]["bytes"] == 4 * per_step + extras
            assert not torch.equal(r0["accum_local_first"], r1["accum_local_first"])           # local so far ...
            assert torch.equal(r0["accum_grad"], r1["accum_grad"])                             # ... averaged sum afterwards
        else:
            assert r0["sync"]["bytes"] == 3 * per_step        # also in "multi": every element exactly once per step


def test_grad_exchange_kernels_match_host_emulation(dev):
    """`ug_grad_pack_bf16` / `ug_grad_sum_shards_bf16` / `ug_grad_unpack_bf16` against the host emulation that
    tests/test_ddp_cpu.py uses for its 2 / 4 / 8-rank error table: bit for bit, ragged length."""
    from test_ddp_cpu import _synthetic_rank_grads, emulate_bf16_fp32acc
    from unigen_hip import ops
    W, n = 8, (1 << 16) + 24
    grads = [g[:n].contiguous() for g in _synthetic_rank_grads(W, n=1 << 17)]
    stride = -(-n // 8) * 8
    shards = torch.zeros(W * stride, dtype=torch.bfloat16, device=dev)
    for r, g in enumerate(grads):
        ops.grad_pack_bf16(g.to(dev), shards[r * stride:r * stride + n], 1.0)
        assert torch.equal(shards[r * stride:r * stride + n].cpu(), g.to(torch.bfloat16))
    out = torch.empty(n, dtype=torch.bfloat16, device=dev)
    ops.grad_sum_shards_bf16(shards, W, stride, out, 1.0 / W)
    back = torch.empty(n, dtype=torch.float32, device=dev)
    ops.grad_unpack_bf16(out, back)
    assert torch.equal(back.cpu(), emulate_bf16_fp32acc(grads))


def test_rccl_world_one_through_the_exchange_path(dev, tmp_path):
    """First contact with RCCL before any multi-GPU box: a one-rank "nccl" process group, the exchange
    forced on (UNIGEN_DDP_FORCE=1), all three wire formats: hooks -> buckets -> side stream -> pack -> ncclAllReduce /
    all-to-all + all-gather -> unpack -> end-of-backward wait.  At world 1 the mean is the gradient itself: fp32 must return it
    bit for bit, the bf16 formats within one bf16 rounding."""
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path[:0] = [os.environ["UG_ROOT"], os.path.join(os.environ["UG_ROOT"], "ml-unigen_amd"), os.path.join(os.environ["UG_ROOT"], "tests")]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from helpers import additive, golden, llm_config_dir
from models import UniGen
from oracle import weights
g = golden("g2_tiny_unigen.pt"); cfg = g["cfg"]; dev = torch.device("cuda:0")
res = {}
for case in ("off", "fp32", "fp32_rsag", "bf16_fp32acc", "bf16", "ug_comm:fp32", "ug_comm:fp32_rsag", "ug_comm:bf16_fp32acc", "ug_comm:bf16", "dense:fp32"):
    reduce = case.split(":")[-1]
    os.environ["UNIGEN_DDP_TRANSPORT"] = "ug_comm" if case.startswith("ug_comm") else "torch"
    os.environ["UNIGEN_DDP_SPARSE_EMBED"] = "0" if case.startswith("dense") else "1"
    os.environ["UNIGEN_DDP_REDUCE"] = "fp32" if reduce == "off" else reduce
    os.environ["UNIGEN_DDP_FORCE"] = "0" if reduce == "off" else "1"
    m = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=312, llm_model_path=llm_config_dir(cfg), codebook_size=20,
               num_vq_tokens=16, device=dev, init_seed=1).train()
    names = [(n, tuple(p.shape)) for n, p in m.llm.named_parameters()]
    m.llm.load_state_dict(weights.synth_llm_state(names, seed=g["weight_seed"]), strict=False)
    _, l1, l2, l3 = m(input_ids=g["input_ids"].to(dev), attention_mask=additive(g["mask_allow"]).to(dev), labels=g["labels"].to(dev), **g["kw"])
    (l1 + l2 + l3).backward()
    torch.cuda.synchronize()
    eng = m.llm.engine
    res[case] = dict(grad=eng.fp.grad.detach().cpu().clone(), bytes=0 if eng.grad_sync is None else eng.grad_sync.bytes_on_wire,
                     backend=None if eng.grad_sync is None else eng.grad_sync.backend,
                     seen=None if eng.grad_sync is None else eng.grad_sync.ranks_seen(),
                     early=None if eng.grad_sync is None else eng.grad_sync.early_embed_handovers)
    eng.grad_sync = None
torch.save(res, sys.argv[2])
dist.destroy_process_group()
"""
    out = str(tmp_path / "rccl1.pt")
    env = dict(os.environ, UG_ROOT=os.path.dirname(HERE), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code, _free_port(), out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = torch.load(out, weights_only=False)
    base = res["off"]["grad"]
    assert res["off"]["bytes"] == 0
    # (two runs of the same backward differ in the last bits: the embedding scatter-add and the dK / dV finish use fp32 atomics)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for pre, backend in (("", "nccl"), ("ug_comm:", "ug_comm(rccl)")):      # torch.distributed's RCCL | the library's own ug_comm_* entry points
        for k in ("fp32", "fp32_rsag", "bf16_fp32acc", "bf16"):
            r = res[pre + k]
            assert r["backend"] == backend and r["bytes"] > 0, (pre + k, r["backend"], r["bytes"])
            assert r["seen"] == 1 and r["early"] == 1, (pre + k, r["seen"], r["early"])      # the head's dense part left early
        assert rel(res[pre + "fp32"]["grad"], base) < 1e-5 and rel(res[pre + "fp32_rsag"]["grad"], base) < 1e-5
        for k in ("bf16_fp32acc", "bf16"):
            g = res[pre + k]["grad"]
            # (the tied table -- the first 333 x 256 elements -- carries the lookups' fp32 rows on top of the bf16-rounded dense part)
            tail = g[333 * 256:]
            assert torch.equal(tail, tail.to(torch.bfloat16).float()) and rel(g, base) < 2.0 ** -8, (pre + k, rel(g, base))
        print(f"[{backend}, world 1] bytes handed to the collectives: " + ", ".join(f"{k} {res[pre + k]['bytes']}" for k in ("fp32", "bf16_fp32acc", "bf16")))
    assert res["ug_comm:fp32"]["bytes"] == res["fp32"]["bytes"] and res["ug_comm:bf16"]["bytes"] == res["bf16"]["bytes"]
    assert res["ug_comm:fp32_rsag"]["bytes"] == res["fp32"]["bytes"]
    assert res["dense:fp32"]["early"] == 0 and rel(res["dense:fp32"]["grad"], base) < 1e-5


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_rccl_two_gpus_every_transport_and_wire_format(dev, tmp_path):
    """the rank-dependent exchanges over REAL RCCL with world > 1 -- both transports (torch.distributed,
    ug_comm_*), all four wire formats, with the early dense hand-over of the tied table and the lookups' all-gather -- against
    the host emulations of tests/test_ddp_cpu.py.  Runs wherever the box has two GPUs; the pool's one-GPU boxes skip it (the
    same branches run there over gloo with two ranks and over RCCL at world 1)."""
    from test_ddp_cpu import emulate_bf16_fp32acc, emulate_bf16_ring
    world, port = 2, _free_port()
    outs = [str(tmp_path / f"rccl2_{r}.pt") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_rccl_worker.py"), str(r), str(world), port, outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-3000:]
    res = [torch.load(o, weights_only=False) for o in outs]
    V, H = 500, 64
    locs = [r["local"] for r in res]
    look = torch.zeros(V, H)
    for r in res:
        look.index_add_(0, r["ids"], r["rows"], alpha=1.0 / world)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for case in res[0]["cases"]:
        c0, c1 = res[0]["cases"][case], res[1]["cases"][case]
        reduce = case.split(":")[1]
        assert c0["seen"] == world and c0["early"] == 1 and c0["lookup_bytes"] == world * (37 + 5) * (8 + 4 * H), (case, c0["seen"], c0["early"])
        assert torch.equal(c0["grad"], c1["grad"]), case                  # replicas bit-identical
        dense = {"fp32": lambda g: torch.stack(g).mean(0), "fp32_rsag": lambda g: torch.stack(g).mean(0),
                 "bf16_fp32acc": emulate_bf16_fp32acc, "bf16": emulate_bf16_ring}[reduce](locs)
        want = dense.clone()
        want[:V * H] += look.reshape(-1)
        got = c0["grad"]
        if reduce == "bf16_fp32acc":                                      # rank-ordered fp32 sums: the emulation bit for bit
            assert torch.equal(got[V * H:], want[V * H:]), case
        tol = 1e-6 if reduce.startswith("fp32") else 4e-3
        assert rel(got, want) < tol, (case, rel(got, want))
        print(f"[RCCL world 2, {case}] rel err vs emulation {rel(got, want):.2e}, {c0['bytes']} + {c0['lookup_bytes']} bytes, backend {c0['backend']}")


def test_bench_runs_end_to_end_with_two_ranks(dev):
    """`python bench.py --gpus 2` with NO launcher around it: bench.py starts torch.distributed.run itself as
    a child process before anything touches the GPU -- the child line is exactly the one the driver uses for N > 1 -- and relays
    the one JSON line of rank 0.  Both ranks on cuda:0 over gloo (RCCL refuses two ranks on one device): every rank must run every
    step that contains the gradient exchange -- including the instrumented roofline steps -- or rank 0 waits for its peers forever.
    The line must prove that the exchange's communicator saw both ranks."""
    import json
    env = dict(os.environ, UNIGEN_DIST_BACKEND="gloo", UNIGEN_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    assert out["roofline"]["launches_per_step"] > 0 and out["cpu_baseline"] is None
    ex = out["exchange"]
    assert out["ranks_seen"] == 2 and ex["ranks_seen"] == 2 and ex["reduce"] == "fp32" and ex["transport"] == "torch"
    # fp32 payload: every element of the flat gradient buffer once per step + the lookups' (id, row) pairs of both ranks
    assert ex["bytes_on_wire_per_step"] >= 4 * 1.5e9 and ex["lookup_bytes_on_wire_per_step"] == 2 * 16 * 771 * (8 + 4 * 1536)
    # round 5: the line explains the exchange by itself -- every bucket of the last timed step with its hand-over time, queueing
    # and collective duration, the exposed wait of the compute stream at the end of backward, and RCCL's own init summary
    # (None on this gloo rehearsal)
    last = ex["last_step"]
    assert last["buckets"] and all(b["collective_ms"] > 0 and b["mb"] > 0 for b in last["buckets"]) and last["exposed_wait_ms"] >= 0
    assert abs(sum(b["mb"] for b in last["buckets"]) * 1e6 - ex["bytes_on_wire_per_step"]) < 0.02 * ex["bytes_on_wire_per_step"]
    assert "rccl" in ex and ex["rccl"] is None
    # round 6: the one N > 1 run also times the bf16-on-the-wire exchange, in the same process, after the block
    # `value` is quoted on; both accounts with their bucket timelines and exposed waits
    modes = ex["modes"]
    assert set(modes) == {"fp32", "bf16_fp32acc"}, modes.keys()
    for name, m in modes.items():
        assert m["samples_per_s"] > 0 and m["last_step"]["buckets"] and m["exposed_wait_ms"] >= 0, (name, m)
    assert abs(modes["fp32"]["samples_per_s"] - out["value"]) < 1e-2 * out["value"]
    # (payload handed to the collectives: 4 bytes per element either way -- one fp32 all-reduce, or a bf16 all-to-all + a bf16
    # all-gather; what differs is the traffic a ring makes of it)
    assert modes["fp32"]["collective"].startswith("all_reduce") and modes["bf16_fp32acc"]["collective"].startswith("all_to_all")
    assert modes["bf16_fp32acc"]["steps"] == 1
