"""SigLIP-so400m vision tower (reference: models/multimodal_encoder/siglip_encoder.py).
Round-1 status: interface only -- the HIP path for the ViT (patch-embed GEMM, LayerNorm, hd-72
attention, tanh-GELU MLP) is scheduled after the stage-1 training path.
Constructing the tower fails loudly rather than silently running a torch fallback."""
import torch.nn as nn


class SigLipVisionTower(nn.Module):
    def __init__(self, vision_tower, vision_tower_cfg=None, freeze=True, delay_load=False):
        super().__init__()
        raise NotImplementedError("SigLipVisionTower: the gfx950 ViT path is not built yet (SURVEY.md §8 a12)")
