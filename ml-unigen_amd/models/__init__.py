"""`models` -- import-compatible with the reference's `models` package (training/train.py:42-43 does
`from models import UniGen, get_mask_chedule` / `from models.model_registry import ...`)."""
from .unigen import UniGen
from .sampling import *  # noqa: F401,F403
from .multimodal_encoder.magvitv2 import VQGANEncoder, VQGANDecoder, LFQuantizer, MAGVITv2
from .multimodal_encoder.siglip_encoder import SigLipVisionTower
