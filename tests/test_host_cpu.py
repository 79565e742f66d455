"""CPU suite, part 2: host-side logic of the product that needs no GPU -- the C ABI library loads and
exports every symbol the header declares, the sampling helpers reproduce the reference's outputs,
config/registry semantics."""
import ctypes
import math
import os
import re
import subprocess

import pytest
import torch

from helpers import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "unigen_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ug_[a-z0-9_]+)\s*\(", text)))


def test_abi_library_exports_every_declared_symbol():
    from unigen_hip import lib
    syms = _header_symbols()
    assert len(syms) >= 30
    L = ctypes.CDLL(lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    # the ctypes signature table covers the same set (no drift between header and binding)
    bound = set(lib.SIGNATURES) | {"ug_last_error"}
    assert set(syms) == bound, (set(syms) ^ bound)
    lib.load()
    assert lib.load().ug_abi_version() == lib.ABI_VERSION == 7


def test_abi_argument_errors_are_reported_not_thrown():
    from unigen_hip import lib
    L = lib.load()
    rc = L.ug_gemm_bf16(0, 0, 80, 0, 0, 80, 0, 0, 8, 16, 16, 68, 0, 0, 0, 0, 0, 0, -1, 0)   # K % 8 != 0, both row-major: rejected before any launch
    assert rc == -1 and b"multiple of 8" in L.ug_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ml-unigen_amd")
    out = subprocess.run(["grep", "-rIl", "-E", r"^\s*(from|import)\s+oracle", pkg], capture_output=True, text=True).stdout
    assert out.strip() == "", out


def test_sampling_helpers_match_reference_outputs():
    from models import sampling
    g = golden("g5_sampling.pt")
    t = g["t"]
    for name in ("cosine", "linear", "pow2", "pow0.5", "sigmoid"):
        assert torch.equal(sampling.get_mask_chedule(name)(t), g["sched_" + name]), name
    m = sampling.mask_by_random_topk(g["mask_len"], g["probs"], 0.7, generator=torch.Generator().manual_seed(6))
    assert torch.equal(m, g["topk_mask"])
    assert torch.equal(sampling.top_k_top_p_filtering(g["filter_in"].clone(), top_k=5), g["filter_k5"])
    assert torch.equal(sampling.top_k_top_p_filtering(g["filter_in"].clone(), top_p=0.8), g["filter_p"])
    assert torch.equal(sampling.gumbel_noise(torch.zeros(2, 5), generator=torch.Generator().manual_seed(7)), g["gumbel"])
    with pytest.raises(ValueError):
        sampling.get_mask_chedule("nope")
    names = ["log", "gumbel_noise", "gumbel_sample", "top_k", "mask_by_random_topk", "cosine_schedule", "linear_schedule",
             "pow", "sigmoid_schedule", "get_mask_chedule", "top_k_top_p_filtering"]
    assert all(hasattr(sampling, n) for n in names)


def test_registry_and_config_semantics():
    from models.model_registry import get_model_creator, MODEL_REGISTRY
    from models import UniGen, MAGVITv2
    assert get_model_creator("unigen") is UniGen and get_model_creator("/ckpt/MAGVITv2-x") is MAGVITv2
    with pytest.raises(ValueError):
        MODEL_REGISTRY.get("resnet")
    from models.modeling_utils import ConfigMixin, ModelMixin, register_to_config

    class M(ModelMixin, ConfigMixin):
        @register_to_config
        def __init__(self, a, b=2, **kw):
            super().__init__()
            self.register_to_config(c=a + b)
    m = M(1)
    assert m.config.a == 1 and m.config.get("b") == 2 and m.config["c"] == 3 and m.c == 3
    assert m.config.get("missing", 7) == 7
    with pytest.raises(AttributeError):
        m.config.a = 5


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from unigen_hip import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.UniGenHipError):
        lib.load()


def test_ops_refuse_cpu_tensors():
    from unigen_hip import ops, lib
    with pytest.raises(lib.UniGenHipError):
        ops.gemm_nt(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))


def test_prepare_inputs_for_mmu_host_logic_matches_reference():
    """The index assembly of UniGen.prepare_inputs_for_mmu (labels, key-validity mask incl. the reference's eos-cursor quirk,
    part-1 ids, sequence lengths in train / eval mode) against what the REAL reference returned (golden G11); the two device
    ops it calls (embedding lookup, mm_projector) are stubbed, the GPU suite checks them."""
    import types
    import torch
    from helpers import golden
    from models.unigen import UniGen
    g = golden("g11_mmu_inputs.pt")
    t = g["template"]
    tmpl = types.SimpleNamespace(text_tokenizer=types.SimpleNamespace(pad_token_id=t["pad_token_id"]), max_seq_len=t["max_seq_len"],
                                 sptids_dict={k: torch.tensor([v]) for k, v in t["sptids"].items()}, ignore_id=t["ignore_id"],
                                 eos_token_id=t["eos_token_id"], task_token_first=t["task_token_first"])
    x = g["mmu_in"]
    for mode in ("train", "eval"):
        for tag in ("nosys", "sys"):
            fake = types.SimpleNamespace(training=(mode == "train"), mm_projector=lambda f: torch.zeros(f.shape[0], f.shape[1], 4),
                                         llm=types.SimpleNamespace(model=types.SimpleNamespace(
                                             embed_tokens=lambda ids: ids[..., None].float().expand(*ids.shape, 4))))
            e, am, lab, p1 = UniGen.prepare_inputs_for_mmu(fake, x["image_feats"], x["spatial_shapes"], x["input_ids"], x["label_ids"],
                                                           tmpl, x["input_ids_system"] if tag == "sys" else None)
            w = g[f"mmu_{mode}_{tag}"]
            assert torch.equal(am, w["attention_mask"]) and torch.equal(lab, w["labels"]) and torch.equal(p1, w["part1"]), (mode, tag)
            assert e.shape[:2] == w["embeddings"].shape[:2]


def test_masking_options_match_reference_golden():
    """G15: data/masking.py's optional branches (reference :20-22 `eval_mask_ratios`, :33-66 `mask_contiguous_region_prob`) draw
    from Python's `random` only, so the drop-in reproduces the REAL reference's rectangles and ratios bit for bit -- host logic,
    no kernel involved."""
    import math
    import random
    import types
    from data.masking import mask_or_random_replace_tokens
    g = torch.load(os.path.join(ROOT, "tests", "golden", "g15_masking_options.pt"), weights_only=False)

    class Node(dict):
        __getattr__ = dict.__getitem__
    for c in g["cases"]:
        cfg = types.SimpleNamespace(training=Node(min_masking_rate=0.0, eval_mask_ratios=[0.25, 0.5, 0.9], mask_contiguous_region_prob=1.0),
                                    model=types.SimpleNamespace(codebook_size=8192))
        random.seed(c["seed"])
        ids, labels, lw, mp = mask_or_random_replace_tokens(c["tokens"], 159866, cfg, lambda t: torch.cos(t * math.pi * 0.5), is_train=False)
        assert lw is None and torch.equal(ids, c["input_ids"]) and torch.equal(labels, c["labels"]) and torch.equal(mp, c["mask_prob"])
        res = int(c["n"] ** 0.5)
        m = (labels != -100).view(-1, res, res)
        for b in range(m.shape[0]):                       # every mask is one filled rectangle
            rows, cols = m[b].any(1).nonzero().flatten(), m[b].any(0).nonzero().flatten()
            assert int(m[b].sum()) == len(rows) * len(cols) and rows[-1] - rows[0] + 1 == len(rows) and cols[-1] - cols[0] + 1 == len(cols)


def test_bench_self_launch_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` without a launcher spawns torch.distributed.run as a child -- but only when the node has a GPU
    per rank: here (no GPU) it must say so and exit non-zero at once, without initialising anything."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UNIGEN_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "one GPU per rank" in (r.stderr + r.stdout)


def test_bench_power_report_reads_the_busy_card_and_survives_a_cut_off_line(tmp_path):
    """bench.py's `power` object (round 4): from the sampler's file -- two cards of which one ran the steps, a running-average ramp at the
    start of the window, samples outside the window, a last line cut off by the terminate -- the report is the busy card's last two
    thirds; without samples (no hwmon here) it is None and the benchmark goes on."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    ps = bench.PowerSampler.__new__(bench.PowerSampler)
    ps.cap, ps.path = 1400.0, str(tmp_path / "samples.txt")
    ps.proc = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(30)"])
    lines = ["%.4f %d %d %d %d" % (99.0 + 0.02 * i, 240000000, 100000000, 40000000, 90000000) for i in range(10)]          # before the window
    for i in range(30):                                                    # card 0 ramps 600 -> 1340 W over the first third; card 1 idles
        w = 600 + min(i, 10) * 74
        lines.append("%.4f %d %d %d %d" % (100.0 + 0.02 * i, w * 1000000, 1950000000, 41000000, 95000000))
    lines.append("100.6100 1340000000 19")                                # cut off mid-write
    open(ps.path, "w").write("\n".join(lines))
    rep = ps.report(100.0, 100.6)
    assert rep["gpus_sampled"] == 1 and rep["samples"] == 20 and rep["cap_w"] == 1400.0
    assert rep["mean_w"] == 1340.0 and rep["max_w"] == 1340.0 and rep["mean_sclk_mhz"] == 1950.0
    assert not os.path.exists(ps.path) and ps.proc.poll() is not None
    none = bench.PowerSampler.__new__(bench.PowerSampler)
    none.proc = None
    assert none.report(0.0, 1.0) is None


def test_gemm_traffic_record_belongs_to_the_shipped_gemm_source():
    """bench.py reports `roofline.traffic` only while profiles/gemm_traffic_current.json carries the sha256 of the gemm_bf16.hip it runs
    (a stale record reads as null): an edit of the GEMM source without `bash tools/refresh_traffic.sh` on the GPU box fails here."""
    import hashlib
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "gemm_traffic_current.json")))
    src = open(os.path.join(ROOT, "ml-unigen_amd", "csrc", "gemm_bf16.hip"), "rb").read()
    assert rec["gemm_src_sha256"] == hashlib.sha256(src).hexdigest()
    assert rec["launches_per_step"] == 255 and 0.5e9 < rec["traffic_bytes_per_launch"] < 2e9


def test_bench_rccl_debug_summary_parses_an_init_log():
    """bench.py's `exchange.rccl`: the summary of RCCL's NCCL_DEBUG=INFO init output -- channels, this rank's
    rings / trees, transports -- from a log of the usual shape; None without a log (the gloo rehearsal)."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    log = """host:101:101 [0] NCCL INFO NCCL version 2.21.5+hip6.3 HEAD:abc
host:101:140 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7
host:101:140 [0] NCCL INFO Ring 00 : 7 -> 0 -> 1
host:101:140 [0] NCCL INFO Ring 01 : 3 -> 0 -> 5
host:101:140 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 5/-1/-1->0->-1
host:101:140 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC
host:101:140 [0] NCCL INFO Connected all rings
host:101:140 [0] NCCL INFO Connected all trees
host:101:140 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 4 p2p channels per peer
"""
    out = bench.rccl_debug_summary(text=log)
    assert out["coll_channels"] == 16 and out["transports"] == ["P2P/IPC"] and len(out["rings"]) == 2 and out["trees"]
    assert out["connected"] == ["Connected all rings", "Connected all trees"] and "2.21.5" in out["version"]
    assert bench.rccl_debug_summary(path="/nonexistent/rccl.log") is None


def test_no_register_spills_in_kernels_with_counted_waits():
    """tools/check_spills.py (also run by __graft_entry__.build()): no kernel of the must-not-spill set -- GEMM, attention, decode,
    convolution kernels, all with hand-counted `s_waitcnt` -- spills registers, the six audited exceptions aside."""
    import subprocess
    import sys
    objs = [f for f in os.listdir(os.path.join(ROOT, "ml-unigen_amd", "csrc")) if f.endswith(".o")]
    if not objs:
        pytest.skip("objects not built (run __graft_entry__.build())")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spills.py")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "0 spilling in the must-not-spill set" in res.stdout


def test_probe_switch_patch_applies_to_the_product_kernels(tmp_path):
    """The timing / ablation / trace switches live in tools/probes/probe_switches.patch, not in the product kernels;
    the patch must keep applying to the shipped sources (tools/probes/build_variant.py depends on it), and the product sources must
    not contain a probe switch."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "ml-unigen_amd", "csrc")
    dst = tmp_path / "ml-unigen_amd" / "csrc"
    dst.mkdir(parents=True)
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            shutil.copy(os.path.join(csrc, f), dst)
            for line in open(os.path.join(csrc, f)):
                code = line.split("//")[0]                   # (comments may name a switch when they point at the patch)
                for sw in ("UG_MFMA_ORDER", "UG_CPOL_", "UG_EPI_NT", "UG_SWP_ABLATE", "UG_SWB_ABLATE", "UG_GEMM_ABLATE", "UG_GEMM_TRACE", "UG_GEMM_R4",
                           "UG_ADAMW_ABLATE", "UG_EW_PROBE", "UG_ADF_ABLATE", "UG_CONV_TRACE"):
                    assert sw not in code, (f, sw, line)
    res = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "tools", "probes", "probe_switches.patch")], cwd=tmp_path,
                         capture_output=True, text=True)
    assert res.returncode == 0 and "FAILED" not in res.stdout, res.stdout[-2000:]
