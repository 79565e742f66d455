"""Config + checkpoint plumbing with the semantics the reference's callers rely on
(reference: models/modeling_utils.py, a vendored copy of diffusers' ModelMixin/ConfigMixin --
no arithmetic lives there).  Only what the hot path's callers touch is provided:

  * `@register_to_config` on `__init__`, `self.register_to_config(**kw)`, `model.config` (attribute
    and `.get()` access, immutable), attribute passthrough `model.mask_token_id` -> config
    (reference modeling_utils.py:126-142; used at training/train.py:260-261),
  * `save_pretrained(dir, ...)` writing `config.json` + `pytorch_model.bin` | `model.safetensors`
    (reference :257-399; called from utils/checkpoint.py:53-59) and `from_pretrained(dir, ...)`
    (reference :401-855; called from training/train.py:241-245) with the same state-dict key names.
"""
import functools
import inspect
import json
import os

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

    # -- checkpoints
    def save_pretrained(self, save_directory, is_main_process=True, save_function=None, safe_serialization=True,
                        variant=None, state_dict=None, **kwargs):
        os.makedirs(save_directory, exist_ok=True)
        if not is_main_process:
            return
        cfg = dict(self.config)
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True, default=str)
        sd = state_dict if state_dict is not None else self.state_dict()
        sd = {k: v.detach().to("cpu").contiguous() for k, v in sd.items()}
        if safe_serialization:
            from safetensors.torch import save_file
            seen, dedup = {}, {}
            for k, v in sd.items():          # safetensors refuses aliased storage (tied lm_head)
                key = (v.data_ptr(), tuple(v.shape))
                if key in seen:
                    v = v.clone()
                seen[key] = k
                dedup[k] = v
            save_file(dedup, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            (save_function or torch.save)(sd, os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, use_safetensors=None, output_loading_info=False,
                        torch_dtype=None, device=None, **kwargs):
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
        model = cls(**init_kw)
        st = os.path.join(path, "model.safetensors")
        pt = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st) and use_safetensors is not False:
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        res = model.load_state_dict(sd, strict=False)
        model.eval()
        if output_loading_info:
            return model, {"missing_keys": list(res.missing_keys), "unexpected_keys": list(res.unexpected_keys),
                           "mismatched_keys": [], "error_msgs": []}
        return model
