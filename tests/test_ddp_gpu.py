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


@pytest.mark.parametrize("reduce", ["bf16", "fp32"])
def test_two_ranks_on_one_gpu_match_single_process(dev, tmp_path, reduce):
    tol = 1e-2 if reduce == "bf16" else 2e-4
    for mode in ("ddp", "bare"):
        single = _run(1, "single" if mode == "ddp" else "bare", tmp_path, reduce)[0]
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
        if mode == "bare":
            # 3 steps + 1 exchanged accumulation step (+ the ordinary parameters, a few KB); the no_sync step moved nothing
            assert r0["no_sync_bytes"] == 0 and r0["accum_bytes"] > 0
            extras = 4 * 4 * (32 * 256 + 256 + 256 * 256 + 256)          # mm_projector gradients, fp32, 4 exchanges
            assert r0["sync"]["bytes"] == 4 * per_step + extras
            assert not torch.equal(r0["accum_local_first"], r1["accum_local_first"])           # local so far ...
            assert torch.equal(r0["accum_grad"], r1["accum_grad"])                             # ... averaged sum afterwards
        else:
            assert r0["sync"]["bytes"] == 3 * per_step


def test_bench_runs_end_to_end_with_two_ranks(dev):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one JSON line from rank 0), with both ranks
    on cuda:0 over gloo (RCCL refuses two ranks on one device): every rank must run every step that contains the gradient
    exchange -- including the instrumented roofline step -- or rank 0 waits for its peers forever."""
    import json
    env = dict(os.environ, UNIGEN_DIST_BACKEND="gloo", UNIGEN_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    assert out["roofline"]["launches_per_step"] > 0 and out["cpu_baseline"] is None
