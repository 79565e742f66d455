"""name -> model class lookup (reference: models/model_registry.py:49-65; substring match on the
lower-cased name, e.g. a checkpoint path containing 'siglip')."""
from . import UniGen, MAGVITv2, SigLipVisionTower

_TABLE = {"magvitv2": MAGVITv2, "siglip": SigLipVisionTower, "unigen": UniGen}


class ModelRegistry:
    def __init__(self, table=None):
        self._registry = dict(table or {})

    def register(self, key, value):
        self._registry[key.lower()] = value

    def update(self, defines):
        self._registry.update(defines)

    def get(self, key):
        k = key.lower()
        if k in self._registry:
            return self._registry[k]
        for name, creator in self._registry.items():
            if name in k:
                return creator
        raise ValueError(f"Unsupported model type: {key}. Supported types: {list(self._registry.keys())}")


MODEL_REGISTRY = ModelRegistry(_TABLE)


def register_model_class(keyword):
    def deco(cls):
        MODEL_REGISTRY.register(keyword, cls)
        return cls
    return deco


register_model_func = register_model_class


def get_model_creator(keyword):
    return MODEL_REGISTRY.get(keyword)


def model_from_name(name):
    return MODEL_REGISTRY.get(name.lower())(name)
