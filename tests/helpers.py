"""Shared test helpers (weights, tiny configs).  Test-side only."""
import json
import os
import tempfile

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


def llm_config_dir(cfg):
    """A directory with the HF-style config.json that UniGen(llm_model_path=...) reads."""
    d = tempfile.mkdtemp(prefix="tinyqwen_")
    c = dict(cfg)
    c.setdefault("model_type", "qwen2")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(c, f)
    return d


def oracle_lm(cfg, seed, std=0.02):
    """CPU oracle model + the synthetic state dict it was loaded with."""
    from oracle import qwen2_ref, weights
    c = qwen2_ref.Qwen2Cfg(**cfg)
    lm = qwen2_ref.RefCausalLM(c)
    names = [(n, tuple(p.shape)) for n, p in lm.named_parameters()]
    sd = weights.synth_llm_state(names, seed=seed, std=std)
    lm.load_state_dict(sd, strict=False)
    return lm, sd


def additive(allow, dtype=torch.float32):
    from oracle.host_ref import to_additive
    return to_additive(allow).to(dtype)
