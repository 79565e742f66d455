"""Config + checkpoint plumbing with the semantics the reference's callers rely on
(reference: models/modeling_utils.py, a vendored copy of diffusers' ModelMixin/ConfigMixin --
no arithmetic lives there).  Only what the hot path's callers touch is provided:

  * `@register_to_config` on `__init__`, `self.register_to_config(**kw)`, `model.config` (attribute
    and `.get()` access, immutable), attribute passthrough `model.mask_token_id` -> config
    (reference modeling_utils.py:126-142; used at training/train.py:260-261),
  * `save_pretrained(dir, ...)` writing `config.json` + `pytorch_model.bin` | `pytorch_model.safetensors`, sharded
    above `max_shard_size` with the reference's index file (reference :257-399; called from
    utils/checkpoint.py:53-59) and `from_pretrained(dir, ...)` (reference :401-855; called from
    training/train.py:241-245) reading the same files -- including directories written by the reference itself
    (tests/golden/ckpt_ref_*) -- with the same state-dict key names.
"""
import functools
import inspect
import json
import os
import re

import torch


class FrozenConfig(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        raise AttributeError("config is frozen; use model.register_to_config(...)")


def register_to_config(init):
    """Decorator: record the constructor's arguments (with defaults) in `self.config`."""

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        names = [n for n, p in sig.parameters.items() if n != "self" and p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL)]
        cfg = {n: p.default for n, p in sig.parameters.items()
               if n != "self" and p.default is not inspect.Parameter.empty}
        cfg.update(dict(zip(names, args)))
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        self.__dict__["_internal_dict"] = FrozenConfig(cfg)
        init(self, *args, **kwargs)

    return wrapper


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kwargs):
        cur = dict(self.__dict__.get("_internal_dict", {}))
        cur.update(kwargs)
        self.__dict__["_internal_dict"] = FrozenConfig(cur)

    @property
    def config(self):
        return self.__dict__.setdefault("_internal_dict", FrozenConfig())


class ModelMixin(torch.nn.Module):
    _supports_gradient_checkpointing = False

    def __init__(self):
        if "_modules" not in self.__dict__:
            super().__init__()

    def __getattr__(self, name):
        cfg = self.__dict__.get("_internal_dict")
        if cfg is not None and name in cfg and name not in self.__dict__:
            return cfg[name]
        return super().__getattr__(name)

    # -- gradient checkpointing toggles (the reference calls these; activations fit in HBM here)
    def enable_gradient_checkpointing(self):
        self.apply(functools.partial(self._set_gradient_checkpointing, value=True))

    def disable_gradient_checkpointing(self):
        self.apply(functools.partial(self._set_gradient_checkpointing, value=False))

    def _set_gradient_checkpointing(self, module, value=False):
        pass

    # -- checkpoints (file names and sharding as reference models/modeling_utils.py:35-36, 94-97, 305-373)
    def save_pretrained(self, save_directory, is_main_process=True, save_function=None, safe_serialization=True,
                        variant=None, max_shard_size="10GB", state_dict=None, **kwargs):
        """`config.json` + `pytorch_model.safetensors` | `pytorch_model.bin`; a state dict larger than `max_shard_size` is
        split greedily in key order into `pytorch_model-0000i-of-0000N.*` with the reference's index file
        (`diffusion_pytorch_model.{safetensors,bin}.index.json`: metadata.total_size + weight_map)."""
        if os.path.isfile(save_directory):
            raise ValueError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        if not is_main_process:
            return
        cfg = {k: v for k, v in dict(self.config).items() if k not in _RUNTIME_ONLY_KWARGS}
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(save_directory, self.config_name), "w", encoding="utf-8") as f:
            f.write(json.dumps(cfg, indent=2, sort_keys=True, default=str) + "\n")
        sd = state_dict if state_dict is not None else self.state_dict()
        host, cpu_sd = {}, {}
        for k, v in sd.items():              # one host copy per distinct tensor: tied names (lm_head / embed_tokens) stay aliased
            key = (v.data_ptr(), tuple(v.shape), tuple(v.stride()))
            if key not in host:
                host[key] = v.detach().to("cpu").contiguous()
            cpu_sd[k] = host[key]
        sd = cpu_sd
        ext = "safetensors" if safe_serialization else "bin"
        stem = WEIGHTS_STEM if variant is None else f"{WEIGHTS_STEM}.{variant}"
        shards = _split_into_shards(sd, _parse_size(max_shard_size))
        for fn in os.listdir(save_directory):                # leftovers of an earlier, differently sharded save
            if _REGEX_SHARD.fullmatch(fn.rsplit(".", 1)[0]) and fn.startswith(WEIGHTS_STEM):
                os.remove(os.path.join(save_directory, fn))
        names = [f"{stem}.{ext}"] if len(shards) == 1 else [f"{stem}-{i + 1:05d}-of-{len(shards):05d}.{ext}" for i in range(len(shards))]
        for fn, keys in zip(names, shards):
            part = {k: sd[k] for k in keys}
            if safe_serialization:
                from safetensors.torch import save_file
                seen = set()
                for k, v in part.items():        # safetensors refuses aliased storage (tied lm_head): write a copy per name
                    key = (v.data_ptr(), tuple(v.shape))
                    if key in seen:
                        part[k] = v.clone()
                    seen.add(key)
                save_file(part, os.path.join(save_directory, fn), metadata={"format": "pt"})
            else:
                (save_function or torch.save)(part, os.path.join(save_directory, fn))
        if len(shards) > 1:
            uniq = {(v.data_ptr(), tuple(v.shape)): v.numel() * v.element_size() for v in sd.values()}    # tied names count once
            index = {"metadata": {"total_size": sum(uniq.values())},
                     "weight_map": {k: fn for fn, keys in zip(names, shards) for k in keys}}
            idx_name = SAFE_WEIGHTS_INDEX_NAME if safe_serialization else WEIGHTS_INDEX_NAME
            with open(os.path.join(save_directory, idx_name), "w", encoding="utf-8") as f:
                f.write(json.dumps(index, indent=2, sort_keys=True) + "\n")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, use_safetensors=None, output_loading_info=False,
                        torch_dtype=None, device=None, **kwargs):
        """Loads what the reference's save_pretrained (or this one) wrote: a sharded set named by an index file, else
        `pytorch_model.safetensors` (also HF's `model.safetensors`), else `pytorch_model.bin`."""
        path = str(pretrained_model_name_or_path)
        with open(os.path.join(path, cls.config_name)) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        cfg.update(kwargs)
        sig = inspect.signature(cls.__init__)
        accepts_kw = any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())
        init_kw = {k: v for k, v in cfg.items() if accepts_kw or k in sig.parameters}
        if device is not None and ("device" in sig.parameters or accepts_kw):
            init_kw["device"] = device
        if accepts_kw:
            init_kw.setdefault("init_seed", -1)              # every weight comes from the files: skip the random init
        model = cls(**init_kw)
        sd = _read_checkpoint(path, use_safetensors)
        res = model.load_state_dict(sd, strict=False)
        model.eval()
        if output_loading_info:
            return model, {"missing_keys": list(res.missing_keys), "unexpected_keys": list(res.unexpected_keys),
                           "mismatched_keys": [], "error_msgs": []}
        return model


CONFIG_NAME = "config.json"
WEIGHTS_STEM = "pytorch_model"
WEIGHTS_NAME = "pytorch_model.bin"                           # reference models/modeling_utils.py:35
SAFETENSORS_WEIGHTS_NAME = "pytorch_model.safetensors"       # :36
WEIGHTS_INDEX_NAME = "diffusion_pytorch_model.bin.index.json"                 # diffusers' constants, used at :376
SAFE_WEIGHTS_INDEX_NAME = "diffusion_pytorch_model.safetensors.index.json"
_INDEX_CANDIDATES = (SAFE_WEIGHTS_INDEX_NAME, WEIGHTS_INDEX_NAME, "pytorch_model.bin.index.json",      # :94-97
                     "pytorch_model.safetensors.index.json", "model.safetensors.index.json")
_RUNTIME_ONLY_KWARGS = ("device", "init_seed")
_REGEX_SHARD = re.compile(r"(.*?)-\d{5}-of-\d{5}")


def _parse_size(size):
    if isinstance(size, int):
        return size
    m = re.fullmatch(r"\s*(\d+(?:\.\d+)?)\s*([KMGT]?i?B)\s*", str(size), re.IGNORECASE)
    if not m:
        raise ValueError(f"max_shard_size must be an int or a string like '5GB' (got {size!r})")
    unit = m.group(2).upper()
    base = 1024 if "I" in unit else 1000
    return int(float(m.group(1)) * base ** {"B": 0, "K": 1, "M": 2, "G": 3, "T": 4}[unit[0]])


def _split_into_shards(sd, limit):
    """Greedy split in key order; tensors sharing storage (the tied lm_head) stay in one shard and count once."""
    shards, cur, size, where = [], [], 0, {}
    for k, v in sd.items():
        key = (v.data_ptr(), tuple(v.shape))
        if key in where:
            shards_idx = where[key]
            (cur if shards_idx == len(shards) else shards[shards_idx]).append(k)
            continue
        nbytes = v.numel() * v.element_size()
        if cur and size + nbytes > limit:
            shards.append(cur)
            cur, size = [], 0
        cur.append(k)
        size += nbytes
        where[key] = len(shards)
    if cur or not shards:
        shards.append(cur)
    return shards


def _load_file(fn):
    if fn.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(fn)
    return torch.load(fn, map_location="cpu", weights_only=True)


def _read_checkpoint(path, use_safetensors=None):
    for name in _INDEX_CANDIDATES:
        idx = os.path.join(path, name)
        if os.path.isfile(idx):
            with open(idx) as f:
                weight_map = json.load(f)["weight_map"]
            sd = {}
            for fn in sorted(set(weight_map.values())):
                sd.update(_load_file(os.path.join(path, fn)))
            missing = [k for k in weight_map if k not in sd]
            if missing:
                raise FileNotFoundError(f"{name} lists tensors absent from the shard files: {missing[:5]}")
            return sd
    cands = [SAFETENSORS_WEIGHTS_NAME, "model.safetensors"] if use_safetensors is not False else []
    cands.append(WEIGHTS_NAME)
    for name in cands:
        fn = os.path.join(path, name)
        if os.path.isfile(fn):
            return _load_file(fn)
    raise FileNotFoundError(f"no checkpoint under {path}: looked for {', '.join(_INDEX_CANDIDATES[:2] + tuple(cands))}")
