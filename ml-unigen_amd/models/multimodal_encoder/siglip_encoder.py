"""SigLIP-so400m vision tower on the gfx950 fp32 kernels -- drop-in for the reference's
`models/multimodal_encoder/siglip_encoder.py:SigLipVisionTower` (understanding branch of the SFT /
DPO / CoT-V configurations).

Semantics restated from the reference: patch-embedding conv 14x14 stride 14 (`SigLipVisionEmbeddings`
:152-178) + learned position embedding; `num_hidden_layers - 1` encoder layers (the tower deletes the
last one, :573) of pre-LayerNorm(eps 1e-6) attention (16 heads x 72, fp32 softmax, :181-243) and a
tanh-GELU MLP (:247-259); the pooling head is replaced by Identity and `forward` returns
`hidden_states[-1]`, i.e. the output of the last kept layer BEFORE `post_layernorm` (:579-590).
The tower is frozen fp32 in every shipped config (it runs outside autocast), so everything here is
fp32 on the f32 matrix cores: linears via `ug_linear_f32` (bias / GELU / residual in the epilogue),
attention as two batched fp32 GEMMs around a row softmax, LayerNorm one wave per row.  torch modules
only HOLD the parameters under the reference checkpoint's names
(`vision_tower.vision_model.encoder.layers.N.self_attn.q_proj.weight`, ...).
"""
import json
import os

import torch
import torch.nn as nn

from unigen_hip import ops
from unigen_hip.lib import UniGenHipError

_SPLIT_LINEAR = os.environ.get("UNIGEN_CONV_FP32_MFMA", "0") != "1"
_UNFUSED_ATTN = os.environ.get("UNIGEN_SIGLIP_UNFUSED_ATTN", "0") == "1"      # GEMM -> softmax -> GEMM with materialised scores (round-1 form, A/B)
_SO400M = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, num_channels=3,
               image_size=384, patch_size=14, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh")


class SigLipVisionConfig:
    model_type = "siglip_vision_model"

    def __init__(self, hidden_size=1152, image_mean=(0.5, 0.5, 0.5), intermediate_size=4304, num_hidden_layers=27,
                 num_attention_heads=16, num_channels=3, image_size=384, patch_size=16, hidden_act="gelu_pytorch_tanh",
                 layer_norm_eps=1e-6, attention_dropout=0.0, **kwargs):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.num_channels, self.patch_size, self.image_size = num_channels, patch_size, image_size
        self.attention_dropout, self.layer_norm_eps, self.hidden_act, self.image_mean = attention_dropout, layer_norm_eps, hidden_act, image_mean
        if hidden_act != "gelu_pytorch_tanh":
            raise UniGenHipError(f"SigLIP activation {hidden_act} is not implemented (so400m uses gelu_pytorch_tanh)")

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        f = os.path.join(str(path), "config.json")
        if os.path.exists(f):
            with open(f) as fh:
                d = json.load(fh)
            if d.get("model_type") == "siglip":
                d = d["vision_config"]
            return cls(**d)
        if "so400m" in str(path).lower():
            return cls(**_SO400M)
        raise UniGenHipError(f"no config.json under {path} and the name is not a known SigLIP variant")


class SigLipImageProcessor:
    """Resize(bicubic) -> rescale 1/255 -> normalise(0.5, 0.5) -> CHW, as the reference's processor (:37-74);
    host-side image I/O, not on the GPU path."""

    def __init__(self, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), size=(384, 384), rescale_factor=1 / 255):
        self.image_mean, self.image_std, self.size, self.rescale_factor = image_mean, image_std, size, rescale_factor

    def preprocess(self, images, return_tensors="pt"):
        import numpy as np
        from PIL import Image
        if isinstance(images, Image.Image):
            images = [images]
        out = []
        for im in images:
            if not isinstance(im, Image.Image):
                im = Image.fromarray(np.asarray(im))
            im = im.convert("RGB").resize((self.size[1], self.size[0]), resample=Image.BICUBIC)
            a = np.asarray(im).astype(np.float32) * self.rescale_factor
            a = (a - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
            out.append(torch.from_numpy(a).permute(2, 0, 1))
        return {"pixel_values": torch.stack(out) if return_tensors == "pt" else out}

    __call__ = preprocess


class _Holder(nn.Module):
    pass


def _encoder_layer(c):
    l = _Holder()
    l.layer_norm1 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
    l.self_attn = _Holder()
    for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
        setattr(l.self_attn, n, nn.Linear(c.hidden_size, c.hidden_size))
    l.layer_norm2 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
    l.mlp = _Holder()
    l.mlp.fc1 = nn.Linear(c.hidden_size, c.intermediate_size)
    l.mlp.fc2 = nn.Linear(c.intermediate_size, c.hidden_size)
    return l


class SigLipVisionModel(nn.Module):
    """Parameter container with the HF layout: vision_model.{embeddings,encoder.layers,post_layernorm}."""

    def __init__(self, c):
        super().__init__()
        vm = _Holder()
        vm.embeddings = _Holder()
        vm.embeddings.patch_embedding = nn.Conv2d(c.num_channels, c.hidden_size, c.patch_size, c.patch_size)
        n_pos = (c.image_size // c.patch_size) ** 2
        vm.embeddings.position_embedding = nn.Embedding(n_pos, c.hidden_size)
        vm.encoder = _Holder()
        vm.encoder.layers = nn.ModuleList([_encoder_layer(c) for _ in range(c.num_hidden_layers)])
        vm.post_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        vm.head = nn.Identity()
        self.vision_model = vm


class SigLipVisionTower(nn.Module):
    def __init__(self, vision_tower, vision_tower_cfg=None, freeze=True, delay_load=False, config=None):
        super().__init__()
        self.is_loaded = False
        self.config = config if config is not None else SigLipVisionConfig.from_pretrained(vision_tower)
        self.vision_tower_name = vision_tower
        self.image_processor = SigLipImageProcessor(size=(self.config.image_size, self.config.image_size))
        self._packed = {}
        if not delay_load:
            self.load_model(freeze)

    # ------------------------------------------------------------------ weights
    def load_model(self, freeze=True, device_map=None):
        if self.is_loaded:
            return
        self.vision_tower = SigLipVisionModel(self.config)
        path = str(self.vision_tower_name)
        files = [f for f in os.listdir(path) if f.endswith(".safetensors")] if os.path.isdir(path) else []
        if files:
            from safetensors.torch import load_file
            sd = {}
            for f in files:
                sd.update(load_file(os.path.join(path, f)))
            sd = {k: v for k, v in sd.items() if k.startswith("vision_model.")}
            self.vision_tower.load_state_dict(sd, strict=False)
        del self.vision_tower.vision_model.encoder.layers[-1:]          # the tower drops the last layer (:573)
        self.vision_tower.requires_grad_(not freeze)
        self.is_loaded = True

    def _apply(self, fn, *a, **k):
        self.__dict__["_packed"] = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.__dict__["_packed"] = {}
        return super().load_state_dict(*a, **k)

    def _pack(self):
        """Once per weight version: NHWC patch-embedding weights (RGB padded to 4 channels), fused q|k|v."""
        vm = self.vision_tower.vision_model
        key = tuple(p._version for p in self.vision_tower.parameters())
        if self._packed.get("key") == key:
            return self._packed
        pe = vm.embeddings.patch_embedding
        w = pe.weight.detach().float()
        if not w.is_cuda:
            raise UniGenHipError("SigLipVisionTower must live in GPU memory; there is no CPU implementation")
        w4 = torch.cat([w, w.new_zeros(w.shape[0], 4 - w.shape[1], *w.shape[2:])], 1)
        wp, cpad = ops.pack_conv_weight(w4)
        pk = {"key": key, "patch_w": wp, "patch_cpad": cpad, "patch_b": pe.bias.detach().float().contiguous(), "qkv": [],
              "split": []}
        for l in vm.encoder.layers:
            a = l.self_attn
            wq = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight]).detach().float().contiguous()
            bq = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]).detach().float().contiguous()
            pk["qkv"].append((wq, bq))
            if _SPLIT_LINEAR:       # fp32-accurate projections on the f16 matrix cores (scaled two-way operand split, three products)
                pk["split"].append(tuple(ops.split_linear_weight(w) for w in
                                         (wq, a.out_proj.weight, l.mlp.fc1.weight, l.mlp.fc2.weight)))
        self._packed = pk
        return pk

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def _encode(self, images):
        c = self.config
        vm = self.vision_tower.vision_model
        pk = self._pack()
        B = images.shape[0]
        D, Hh = c.hidden_size, c.num_attention_heads
        hd = D // Hh
        g = c.image_size // c.patch_size
        T = g * g
        x = ops.nchw_to_nhwc(images.float().contiguous(), 4)
        pos = vm.embeddings.position_embedding.weight.detach().float()
        pos_b = pos.unsqueeze(0).expand(B, T, D).contiguous().view(B, g, g, D)
        h = ops.conv2d_nhwc(x, pk["patch_w"], pk["patch_cpad"], pk["patch_b"], D, c.patch_size, stride=c.patch_size, pad=0,
                            residual=pos_b).view(B * T, D)
        ldS = ops.round_up(T, 4)
        scale = float(hd) ** -0.5
        # q|k|v of every token, reused by all layers.  Four slack rows that stay zero: the P.V contraction runs over
        # T rounded up to four keys (16-byte loads); the padded probabilities are zero and the rows they meet are finite
        qkv = torch.empty((B * T + 4, 3 * D), dtype=torch.float32, device=h.device)
        qkv[B * T:].zero_()
        Tp = ops.round_up(T, 4)
        for li, l in enumerate(vm.encoder.layers):
            xn = ops.layernorm_f32(h, l.layer_norm1.weight.detach(), l.layer_norm1.bias.detach(), c.layer_norm_eps)
            wq, bq = pk["qkv"][li]
            sp = pk["split"][li] if pk["split"] else None

            def lin(x, j, W, bias, **kw):
                if sp is not None:
                    return ops.linear_split(x, sp[j][0], sp[j][1], W.shape[0], bias, **kw)
                return ops.linear_f32(x, W, bias, **kw)

            lin(xn, 0, wq, bq, out=qkv, M=B * T)
            ctx = torch.empty((B * T, D), dtype=torch.float32, device=h.device)
            if hd % 4 == 0 and hd <= 80 and not _UNFUSED_ATTN:
                # one flash-style kernel per layer: scores stay in registers, both contractions on the split-f16 path
                ops.siglip_attn(qkv, ctx, B, T, Hh, hd, scale)
            else:
                # all heads of all images per launch: batch = (head, image); scores [B, Hh, T, ldS]
                s = torch.empty((B, Hh, T, ldS), dtype=torch.float32, device=h.device)
                ops.gemm_f32_nested(qkv[:, 0:D], qkv[:, D:2 * D], s, b_is_nk=True, M=T, N=T, K=hd, batch_in=Hh, batch_out=B,
                                    lda=3 * D, ldb=3 * D, ldc=ldS, sa=(hd, T * 3 * D), sb=(hd, T * 3 * D), sc=(T * ldS, Hh * T * ldS))
                ops.softmax_rows_(s.view(B * Hh * T, ldS), scale, cols=T)
                ops.gemm_f32_nested(s, qkv[:, 2 * D:], ctx, b_is_nk=False, M=T, N=hd, K=Tp, batch_in=Hh, batch_out=B, lda=ldS,
                                    ldb=3 * D, ldc=D, sa=(T * ldS, Hh * T * ldS), sb=(hd, T * 3 * D), sc=(hd, T * D))
            o = l.self_attn.out_proj
            h = lin(ctx, 1, o.weight.detach(), o.bias.detach(), residual=h)
            xn2 = ops.layernorm_f32(h, l.layer_norm2.weight.detach(), l.layer_norm2.bias.detach(), c.layer_norm_eps)
            m = lin(xn2, 2, l.mlp.fc1.weight.detach(), l.mlp.fc1.bias.detach(), act=1)
            h = lin(m, 3, l.mlp.fc2.weight.detach(), l.mlp.fc2.bias.detach(), residual=h)
        return h.view(B, T, D)

    def forward(self, images):
        # The tower has no backward here: every shipped config freezes it (configs/*: model.vision_tower.freeze true).
        # Training it (mm_tunable_parts containing 'mm_vision_tower', train_w_clip_vit.py:311-312) must not silently
        # train nothing.
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.vision_tower.parameters()):
            raise UniGenHipError("SigLipVisionTower: training the vision tower is not implemented (forward only); freeze it "
                                 "(`vision_tower.requires_grad_(False)`, config model.vision_tower.freeze) or run under no_grad")
        if type(images) is list:
            return [self._encode(im.to(device=self.device, dtype=self.dtype).unsqueeze(0)).to(im.dtype) for im in images]
        return self._encode(images.to(device=self.device, dtype=self.dtype)).to(images.dtype)

    # ------------------------------------------------------------------ reference properties
    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return next(self.vision_tower.parameters()).dtype

    @property
    def device(self):
        return next(self.vision_tower.parameters()).device

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def image_size(self):
        return self.config.image_size
