"""Vision-tower factory (reference: models/multimodal_encoder/builder.py:9-15)."""
from .siglip_encoder import SigLipVisionTower


def get_vision_tower(model_name, freeze=True):
    if "siglip2" in model_name and "naflex" in model_name:
        raise ValueError("SigLIP-2 NaFlex towers are not used by any shipped UniGen config and are not implemented")
    if "siglip" in model_name:
        return SigLipVisionTower(model_name, freeze=freeze)
    raise ValueError(f"model_type {model_name} not supported.")
