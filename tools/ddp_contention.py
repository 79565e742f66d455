"""CU-contention model of the data-parallel gradient exchange on ONE GPU.

The headline step's backward issues one-round GEMM grids that assume all 256 CUs; on 8 GPUs an RCCL ring all-reduce holds G
workgroups (one per channel) on the side stream for as long as a bucket is on the links.  This tool runs bench.py's step with
the REAL exchange machinery armed at world size 1 (UNIGEN_DDP_FORCE=1: hooks -> buckets -> side stream -> end-of-backward
wait) but replaces each bucket's collective with a copy kernel of exactly G workgroups that moves the bucket's ring traffic
(2 x 7/8 x payload bytes) through HBM -- a G-workgroup copy runs at ~G x 25 GB/s, i.e. G = 12..16 is the ~300-400 GB/s of
7 xGMI links -- and reports ms per step against G and the wire format.  t(G = 0) / t(G) is the predicted weak-scaling
efficiency of the compute side (link latency and RCCL's own protocol overhead not included).

    python tools/ddp_contention.py            # builds nothing: tools/probes/_build/occupy.so travels with the tree"""
import ctypes
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
os.environ["UNIGEN_DDP_FORCE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import bench
from models import MAGVITv2, UniGen
from unigen_hip import ops
from unigen_hip.ddp import FlatGradSync
from unigen_hip.optim import FusedAdamW

occ = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "_build", "occupy.so"))
occ.occupy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
STATE = {"groups": 0, "bytes_per_elem": 4, "scratch": None, "moved": 0}


def fake_flush_piece(self, lo, hi):
    """the bucket's collective as a G-workgroup copy of its ring traffic on the exchange's side stream"""
    n = hi - lo
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    with torch.cuda.stream(self.stream):
        self.stream.wait_event(ev)
        if STATE["groups"] > 0:
            traffic = int(n * STATE["bytes_per_elem"] * 2 * 7 / 8)          # reduce-scatter + all-gather of a ring over 8 ranks
            half = traffic // 2 // 16 * 16                                   # the copy reads `half` and writes `half`
            buf = STATE["scratch"]
            occ.occupy_launch(buf.data_ptr(), buf.data_ptr() + buf.numel() * 2, half, STATE["groups"], 1, torch.cuda.current_stream().cuda_stream)
            STATE["moved"] += traffic


def main(steps=8):
    dist.init_process_group("gloo", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    model = UniGen(w_und_encoder=False, vocab_size=bench.VOCAB, llm_vocab_size=bench.TEXT_VOCAB, llm_model_path="Qwen2.5-1.5B-Instruct",
                   codebook_size=bench.CODEBOOK, num_vq_tokens=bench.NVQ, load_from_pretrained=True, device=dev, init_seed=-1)
    model.llm.init_weights_device(bench.SEED)
    model.train()
    vq = MAGVITv2().to(dev).eval().requires_grad_(False)
    bench.init_magvit_device(vq, bench.SEED)
    opt = FusedAdamW([{"params": [p for p in model.parameters()], "weight_decay": 0.01}], lr=1e-4, overlap=True)
    B, T = 16, 511
    g = torch.Generator(device=dev).manual_seed(1)
    images = torch.rand(B, 3, 256, 256, device=dev, generator=g) * 2 - 1
    text = torch.randint(0, 151643, (B, T), device=dev, generator=g)
    STATE["scratch"] = torch.empty(1 << 30, dtype=torch.int32, device=dev)      # 4 GiB: source half | destination half
    FlatGradSync._flush_piece = fake_flush_piece

    def step():
        codes = vq.get_code(images) + bench.TEXT_VOCAB
        ids, labels, mask = bench.t2i_rows(ops, text, torch.full_like(codes, bench.MASK_ID), codes)
        _, l, _, _ = model(input_ids=ids, attention_mask=mask, labels=labels, batch_size_t2i=B, max_seq_length=T + 1, num_vq_tokens=bench.NVQ)
        l.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    rows = []
    for fmt, bpe in (("fp32", 4), ("bf16", 2)):
        for groups in (0, 8, 16, 32, 64):
            STATE.update(groups=groups, bytes_per_elem=bpe, moved=0)
            sampler = bench.PowerSampler()              # round 5: the copy workgroups draw watts the GEMMs lose -- package W / MHz per row
            step()
            torch.cuda.synchronize()
            w0, t0 = time.time(), time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            pw = sampler.report(w0, time.time()) or {}
            rows.append({"wire": fmt, "workgroups": groups, "ms_per_step": round(ms, 2), "GB_per_step": round(STATE["moved"] / (steps + 1) / 1e9, 2),
                         "package_w": pw.get("mean_w"), "sclk_mhz": pw.get("mean_sclk_mhz")})
            print(json.dumps(rows[-1]), flush=True)
            if groups == 0 and fmt == "bf16":
                pass
    # rate of the G-workgroup copy alone (what link bandwidth it stands for)
    buf = STATE["scratch"]
    for groups in (8, 16, 32, 64):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        occ.occupy_launch(buf.data_ptr(), buf.data_ptr() + buf.numel() * 2, 1 << 30, groups, 1, torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"copy_alone_workgroups": groups, "GB_per_s_read_plus_write": round(2 * (1 << 30) / e0.elapsed_time(e1) / 1e6, 1)}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
