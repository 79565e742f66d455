"""ORACLE / test infrastructure: deterministic synthetic weights, independent of module
construction order (each tensor has its own generator seeded from crc32(name) ^ seed), so the same
state dict can be rebuilt in the build container (to drive the real reference when generating
golden vectors) and on the GPU box (to load into the HIP model and the CPU restatement)."""
import math
import zlib

import torch


def _gen(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_llm_state(named_shapes, seed, std=0.02):
    """HF-Qwen2-style init (normal(0, std) matrices) but with non-trivial biases / norm gains so
    every parameter's gradient path is exercised.  named_shapes: iterable of (name, shape)."""
    out = {}
    for name, shape in named_shapes:
        g = _gen(name, seed)
        if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            out[name] = 0.02 * torch.randn(shape, generator=g)
        else:
            out[name] = std * torch.randn(shape, generator=g)
    return out


def synth_magvit_state(named_shapes, seed):
    """Fan-in scaled conv weights, mild GroupNorm affine: keeps activations O(1) through ~50 convs."""
    out = {}
    for name, shape in named_shapes:
        g = _gen(name, seed)
        if name.startswith("quantize."):
            continue
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            out[name] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        elif "norm" in name and name.endswith(".weight"):
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            out[name] = 0.05 * torch.randn(shape, generator=g)
    return out


def synth_images(batch, res, seed):
    """U(-1,1) images, the range the reference's loader normalises to (training/data_loader.py:161-166)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, 3, res, res, generator=g) * 2 - 1


def synth_siglip_state(named_shapes, seed):
    """Fan-in scaled projections, mild LayerNorm affine, small biases / position embeddings."""
    out = {}
    for name, shape in named_shapes:
        g = _gen(name, seed)
        if "layer_norm" in name or "layernorm" in name:
            out[name] = (1.0 + 0.1 * torch.randn(shape, generator=g)) if name.endswith("weight") else 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            out[name] = 0.05 * torch.randn(shape, generator=g)
        elif "position_embedding" in name:
            out[name] = 0.2 * torch.randn(shape, generator=g)
        else:
            fan_in = math.prod(shape[1:])
            out[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return out
