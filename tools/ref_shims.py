"""Import shims that let the reference's `models` / `training.prompting_utils` / `data.masking` be
imported in THIS container (they need diffusers, jaxtyping, typeguard, omegaconf, torchvision at
import time, none of which is installed -- SURVEY.md §8c).  Used only by tools/make_golden.py to
generate fixtures; nothing here ships to the GPU box as part of the product or the tests.
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import tempfile
import types

REFERENCE = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Frozen(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def install():
    import functools
    import inspect
    import logging

    import torch

    # ---- diffusers (only names; ConfigMixin/register_to_config carry real semantics)
    class ConfigMixin:
        config_name = "config.json"

        def register_to_config(self, **kw):
            d = dict(self.__dict__.get("_internal_dict", {}))
            d.update(kw)
            self.__dict__["_internal_dict"] = _Frozen(d)

        @property
        def config(self):
            return self._internal_dict

        def save_config(self, save_directory, **kw):
            """what diffusers' ConfigMixin.save_config / to_json_string write: the registered kwargs plus _class_name and
            _diffusers_version, sorted keys, indent 2"""
            os.makedirs(save_directory, exist_ok=True)
            cfg = dict(self._internal_dict)
            cfg["_class_name"] = type(self).__name__
            cfg["_diffusers_version"] = "0.0.0-shim"
            with open(os.path.join(save_directory, self.config_name), "w", encoding="utf-8") as f:
                f.write(json.dumps(cfg, indent=2, sort_keys=True) + "\n")

    def register_to_config(init):
        @functools.wraps(init)
        def inner(self, *a, **kw):
            sig = inspect.signature(init)
            cfg = {n: p.default for n, p in sig.parameters.items() if n != "self" and p.default is not inspect._empty}
            names = [n for n in sig.parameters if n != "self"]
            cfg.update(dict(zip(names, a)))
            cfg.update(kw)
            self.__dict__["_internal_dict"] = _Frozen(cfg)     # recorded BEFORE __init__ runs (it reads self.config)
            init(self, *a, **kw)
        return inner

    d = _mod("diffusers", __version__="0.0.0-shim")
    d.__path__ = []
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    lg = types.SimpleNamespace(get_logger=lambda name=None: logging.getLogger(name or "diffusers"))
    _mod("diffusers.utils", FLAX_WEIGHTS_NAME="flax.msgpack", SAFE_WEIGHTS_INDEX_NAME="diffusion_pytorch_model.safetensors.index.json",
         WEIGHTS_INDEX_NAME="diffusion_pytorch_model.bin.index.json",      # diffusers.utils.constants values
         _add_variant=lambda n, v=None: n, _get_checkpoint_shard_files=None, _get_model_file=None,
         deprecate=lambda *a, **k: None, is_accelerate_available=lambda: False,
         is_torch_version=lambda *a: True, logging=lg).__path__ = []
    _mod("diffusers.utils.hub_utils", PushToHubMixin=type("PushToHubMixin", (), {}),
         load_or_create_model_card=None, populate_model_card=None)
    _mod("diffusers.models").__path__ = []
    _mod("diffusers.models.model_loading_utils", _determine_device_map=None, _fetch_index_file=None,
         _load_state_dict_into_model=None, load_model_dict_into_meta=None, load_state_dict=None)

    # ---- typing helpers
    class _Any:
        def __getitem__(self, item):
            return object
    names = "Bool Complex Float Inexact Int Integer Num Shaped UInt".split()
    _mod("jaxtyping", **{n: _Any() for n in names})
    _mod("typeguard", typechecked=lambda f=None, **k: f if f is not None else (lambda g: g))
    _mod("omegaconf", OmegaConf=type("OmegaConf", (), {}), DictConfig=dict)

    # ---- siglip2 tower needs torchvision: stub the module
    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        target = "models.multimodal_encoder.siglip2_encoder"

        def find_spec(self, fullname, path, target=None):
            if fullname == self.target:
                return importlib.machinery.ModuleSpec(fullname, self)
            return None

        def create_module(self, spec):
            return None

        def exec_module(self, module):
            module.SigLip2VisionTower = None
    sys.meta_path.insert(0, _Finder())

    torch.Tensor.cuda = lambda self, *a, **k: self      # models/unigen.py:470 hard-codes .cuda()
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)


def write_llm_config_dir(cfg_dict):
    """A directory AutoConfig.from_pretrained can read (models/unigen.py:52)."""
    d = tempfile.mkdtemp(prefix="tinyqwen_")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg_dict, f)
    return d


class _Anything:
    """Attribute sink for modules the reference imports at file scope but the captured function never touches."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


def import_with_stubs(module_name, max_stubs=40):
    """Import a reference module whose file-scope imports need packages this container lacks (wandb, omegaconf, ...):
    every missing top-level package is replaced by a names-only stub and the import retried.  Only used to reach plain
    functions (e.g. training/train_dpo.py:get_batch_logps) whose own body needs none of them."""
    import importlib
    for _ in range(max_stubs):
        try:
            return importlib.import_module(module_name)
        except ModuleNotFoundError as e:
            missing = e.name
            if missing is None or missing.split(".")[0] in ("training", "models"):
                raise
            parts = missing.split(".")
            import importlib.machinery
            for i in range(1, len(parts) + 1):
                nm = ".".join(parts[:i])
                if nm not in sys.modules:
                    st = _StubModule(nm)
                    st.__spec__ = importlib.machinery.ModuleSpec(nm, None)
                    sys.modules[nm] = st
        except ImportError as e:          # a names-only shim installed earlier lacks this name: add it
            import re
            m = re.match(r"cannot import name '(\w+)' from '([\w.]+)'", str(e))
            if m and m.group(2) in sys.modules:
                setattr(sys.modules[m.group(2)], m.group(1), _Anything())
            else:
                raise
    raise RuntimeError(f"too many missing modules importing {module_name}")
