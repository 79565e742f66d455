"""SigLIP-so400m vision tower on the gfx950 fp32 kernels -- drop-in for the reference's
`models/multimodal_encoder/siglip_encoder.py:SigLipVisionTower` (understanding branch of the SFT /
DPO / CoT-V configurations).

Semantics restated from the reference: patch-embedding conv 14x14 stride 14 (`SigLipVisionEmbeddings`
:152-178) + learned position embedding; `num_hidden_layers - 1` encoder layers (the tower deletes the
last one, :573) of pre-LayerNorm(eps 1e-6) attention (16 heads x 72, fp32 softmax, :181-243) and a
tanh-GELU MLP (:247-259); the pooling head is replaced by Identity and `forward` returns
`hidden_states[-1]`, i.e. the output of the last kept layer BEFORE `post_layernorm` (:579-590).
The tower is frozen fp32 in every shipped config (it runs outside autocast), so everything here is
fp32: linears on the fp32-accurate split-f16 contraction (`ug_linear_split`; `ug_linear_f32` with UNIGEN_CONV_FP32_MFMA=1),
one flash-style fp32 attention kernel, LayerNorm one wave per row.  An UNFROZEN tower (reference models/unigen.py:111
`freeze=False`) trains through a hand-written fp32 backward (`_TowerFn` / `_backward`, csrc/siglip_bwd.hip).  torch modules
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


class _TowerFn(torch.autograd.Function):
    """SigLipVisionTower.forward for a tower with trainable parameters: forward = `_encode` keeping per-layer activations,
    backward = `_backward` (hand-written, fp32).  The parameters are inputs so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, tower, images, *params):
        saved = []
        out = tower._encode(images, save=saved)
        ctx.tower, ctx.images, ctx.saved, ctx.params = tower, images, saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = ctx.tower._backward(ctx.images, ctx.saved, dout.contiguous())
        ctx.saved = None
        return (None, None) + tuple(grads[p].to(p.dtype) if p in grads else None for p in ctx.params)


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
    def _encode(self, images, save=None):
        """save: a list that receives, per encoder layer, the tensors the backward of an UNFROZEN tower needs (layer input, q|k|v,
        attention context, post-attention residual stream, fc1 pre-activation); None = inference (nothing is kept)."""
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
        train = save is not None
        ldS = ops.round_up(T, 4)
        scale = float(hd) ** -0.5
        # q|k|v of every token, reused by all layers.  Four slack rows that stay zero: the P.V contraction runs over
        # T rounded up to four keys (16-byte loads); the padded probabilities are zero and the rows they meet are finite
        qkv = torch.empty((B * T + 4, 3 * D), dtype=torch.float32, device=h.device)
        qkv[B * T:].zero_()
        Tp = ops.round_up(T, 4)
        for li, l in enumerate(vm.encoder.layers):
            if train and li > 0:                      # every layer keeps its own q|k|v when a backward follows
                qkv = torch.empty((B * T + 4, 3 * D), dtype=torch.float32, device=h.device)
                qkv[B * T:].zero_()
            h_in = h
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
            if train:                                 # the GELU's derivative needs the pre-activation: activation as its own pass
                pre = lin(xn2, 2, l.mlp.fc1.weight.detach(), l.mlp.fc1.bias.detach())
                m = ops.gelu_tanh_f32(pre)
                save.append((h_in, qkv, ctx, h, pre))
            else:
                m = lin(xn2, 2, l.mlp.fc1.weight.detach(), l.mlp.fc1.bias.detach(), act=1)
            h = lin(m, 3, l.mlp.fc2.weight.detach(), l.mlp.fc2.bias.detach(), residual=h)
        return h.view(B, T, D)

    # ------------------------------------------------------------------ backward (unfrozen tower)
    @torch.no_grad()
    def _backward(self, images, saved, dout):
        """Gradient of every tower parameter for d(loss)/d(output) = dout [B, T, D] (reference: autograd through
        SigLipVisionModel when `freeze=False`, models/unigen.py:111, training/train_w_clip_vit.py:282,311-312).  fp32; every
        contraction on the exact fp32 MFMA GEMM with its operands brought into [rows][k] form by `ug_transpose_f32`;
        LayerNorm / GELU / softmax derivatives in csrc/siglip_bwd.hip; scores are recomputed per layer (not stored).
        -> {parameter: gradient tensor}."""
        c = self.config
        vm = self.vision_tower.vision_model
        B = images.shape[0]
        D, Hh, I = c.hidden_size, c.num_attention_heads, c.intermediate_size
        hd = D // Hh
        g = c.image_size // c.patch_size
        T = g * g
        M = B * T
        Tp = ops.round_up(T, 4)
        scale = float(hd) ** -0.5
        dev = dout.device
        eps = c.layer_norm_eps
        grads = {}
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)

        def mm(a, bt, Mr, N, K, lda=None, ldb=None):
            """a [Mr, K] (row stride lda) @ bt[N, K]^T (row stride ldb) -> [Mr, N]; K a multiple of 4 (zero-padded operands)"""
            return ops.gemm_f32(a, bt, b_is_nk=True, M=Mr, N=N, K=K, lda=a.stride(0) if lda is None else lda,
                                ldb=bt.stride(0) if ldb is None else ldb)[0]

        def linear_bwd(dy, x, W, n_out, n_in):
            """y = x W^T + b -> (dx, dW, db); dy [M, n_out], x [M, n_in]"""
            dyT = ops.transpose_f32(dy, M, n_out)[0]                       # [n_out, Mp], zero tail
            xT = ops.transpose_f32(x, M, n_in)[0]                          # [n_in, Mp]
            dW = mm(dyT, xT, n_out, n_in, dyT.shape[1])
            db = ops.colsum_f32_(dy, z(n_out))
            WT = ops.transpose_f32(W.detach().float().contiguous(), n_out, n_in)[0]     # [n_in, n_out_p]
            dyp = dy
            if n_out % 4:                                                  # contraction over n_out: zero-padded copy of dy
                dyp = z(M, WT.shape[1])
                dyp[:, :n_out] = dy
            dx = mm(dyp, WT, M, n_in, WT.shape[1])
            return dx, dW, db

        dh = dout.reshape(M, D).float().contiguous()
        for li in reversed(range(len(vm.encoder.layers))):
            l = vm.encoder.layers[li]
            h_in, qkv, ctx, h_mid, pre = saved[li]
            a = l.self_attn
            # ---- MLP: h_out = h_mid + fc2(gelu(fc1(LN2(h_mid))))
            m = ops.gelu_tanh_f32(pre)
            dm, grads[l.mlp.fc2.weight], grads[l.mlp.fc2.bias] = linear_bwd(dh, m, l.mlp.fc2.weight, D, I)
            dpre = ops.gelu_tanh_f32(pre, dm)
            xn2 = ops.layernorm_f32(h_mid, l.layer_norm2.weight.detach(), l.layer_norm2.bias.detach(), eps)
            dxn2, grads[l.mlp.fc1.weight], grads[l.mlp.fc1.bias] = linear_bwd(dpre, xn2, l.mlp.fc1.weight, I, D)
            grads[l.layer_norm2.weight], grads[l.layer_norm2.bias] = z(D), z(D)
            dh_mid = ops.layernorm_bwd_f32(dxn2, h_mid, l.layer_norm2.weight.detach().float(), eps, grads[l.layer_norm2.weight],
                                           grads[l.layer_norm2.bias], dres_in=dh)
            # ---- attention output projection: h_mid = h_in + out_proj(ctx)
            dctx, grads[a.out_proj.weight], grads[a.out_proj.bias] = linear_bwd(dh_mid, ctx, a.out_proj.weight, D, D)
            # ---- attention core, all heads of all images per launch: batch = (head, image)
            ldS = ops.round_up(T, 4)
            P = torch.empty((B, Hh, T, ldS), dtype=torch.float32, device=dev)
            nest = dict(batch_in=Hh, batch_out=B)
            ops.gemm_f32_nested(qkv[:, 0:D], qkv[:, D:2 * D], P, b_is_nk=True, M=T, N=T, K=hd, lda=3 * D, ldb=3 * D, ldc=ldS,
                                sa=(hd, T * 3 * D), sb=(hd, T * 3 * D), sc=(T * ldS, Hh * T * ldS), **nest)
            ops.softmax_rows_(P.view(B * Hh * T, ldS), scale, cols=T)
            dP = torch.empty_like(P)                                       # dP = dctx_h V_h^T
            ops.gemm_f32_nested(dctx, qkv[:, 2 * D:], dP, b_is_nk=True, M=T, N=T, K=hd, lda=D, ldb=3 * D, ldc=ldS,
                                sa=(hd, T * D), sb=(hd, T * 3 * D), sc=(T * ldS, Hh * T * ldS), **nest)
            PT = ops.transpose_f32(P, T, T, batch=B * Hh, ld_in=ldS, stride_in=T * ldS)                  # [B*Hh, Tk, Tp(q)]
            ops.softmax_bwd_rows_(P.view(B * Hh * T, ldS), dP.view(B * Hh * T, ldS), scale, T)           # dP becomes dS (x scale)
            dST = ops.transpose_f32(dP, T, T, batch=B * Hh, ld_in=ldS, stride_in=T * ldS)
            dqkv = torch.empty((M, 3 * D), dtype=torch.float32, device=dev)
            # per-head transposes of q, k and dctx ([T, hd] -> [hd, Tp]) as the [N][K] operands of the three contractions
            qT = torch.empty((B, Hh, hd, Tp), dtype=torch.float32, device=dev)
            kT, dcT = torch.empty_like(qT), torch.empty_like(qT)
            for b in range(B):
                qT[b] = ops.transpose_f32(qkv[b * T:, 0:D], T, hd, batch=Hh, ld_in=3 * D, stride_in=hd)
                kT[b] = ops.transpose_f32(qkv[b * T:, D:2 * D], T, hd, batch=Hh, ld_in=3 * D, stride_in=hd)
                dcT[b] = ops.transpose_f32(dctx[b * T:], T, hd, batch=Hh, ld_in=D, stride_in=hd)
            tk = dict(M=T, N=hd, K=Tp, ldb=Tp, ldc=3 * D, sb=(hd * Tp, Hh * hd * Tp), sc=(hd, T * 3 * D), **nest)
            ops.gemm_f32_nested(dP, kT, dqkv[:, 0:D], b_is_nk=True, lda=ldS, sa=(T * ldS, Hh * T * ldS), **tk)          # dQ = dS K
            ops.gemm_f32_nested(dST, qT, dqkv[:, D:2 * D], b_is_nk=True, lda=Tp, sa=(T * Tp, Hh * T * Tp), **tk)        # dK = dS^T Q
            ops.gemm_f32_nested(PT, dcT, dqkv[:, 2 * D:], b_is_nk=True, lda=Tp, sa=(T * Tp, Hh * T * Tp), **tk)         # dV = P^T dctx
            del P, dP, PT, dST
            # ---- fused q|k|v projection and the first LayerNorm
            xn = ops.layernorm_f32(h_in, l.layer_norm1.weight.detach(), l.layer_norm1.bias.detach(), eps)
            wq = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight]).detach().float().contiguous()
            dxn, dWq, dbq = linear_bwd(dqkv, xn, wq, 3 * D, D)
            for j, proj in enumerate((a.q_proj, a.k_proj, a.v_proj)):
                grads[proj.weight], grads[proj.bias] = dWq[j * D:(j + 1) * D], dbq[j * D:(j + 1) * D]
            grads[l.layer_norm1.weight], grads[l.layer_norm1.bias] = z(D), z(D)
            dh = ops.layernorm_bwd_f32(dxn, h_in, l.layer_norm1.weight.detach().float(), eps, grads[l.layer_norm1.weight],
                                       grads[l.layer_norm1.bias], dres_in=dh_mid)
        # ---- embeddings: h0 = conv14x14/14(images) + position embedding (non-overlapping patches: a plain GEMM over [B*T, 3*p*p])
        pe, ps = vm.embeddings.patch_embedding, c.patch_size
        grads[vm.embeddings.position_embedding.weight] = ops.colsum_f32_(dh.view(B, T * D), z(T * D)).view(T, D)
        grads[pe.bias] = ops.colsum_f32_(dh, z(D))
        # (a stride-14 convolution without padding never reads the last image_size - g * 14 rows / columns: 384 -> 378)
        patches = images.float()[:, :, :g * ps, :g * ps].reshape(B, c.num_channels, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(
            M, c.num_channels * ps * ps)
        dhT = ops.transpose_f32(dh, M, D)[0]
        pT = ops.transpose_f32(patches.contiguous(), M, patches.shape[1])[0]
        grads[pe.weight] = mm(dhT, pT, D, patches.shape[1], dhT.shape[1]).view(pe.weight.shape)
        return grads

    def forward(self, images):
        if type(images) is list:
            return [self._forward_one(im.to(device=self.device, dtype=self.dtype).unsqueeze(0)).to(im.dtype) for im in images]
        return self._forward_one(images.to(device=self.device, dtype=self.dtype)).to(images.dtype)

    def _forward_one(self, images):
        # Frozen (every shipped config: model.vision_tower.freeze true) or under no_grad: inference kernels, nothing kept.
        # Unfrozen (mm_tunable_parts containing 'mm_vision_tower', train_w_clip_vit.py:311-312): the same forward with the
        # activations a backward needs, on the autograd graph through _TowerFn.
        params = [p for p in self.vision_tower.parameters() if p.requires_grad]
        if torch.is_grad_enabled() and params:
            return _TowerFn.apply(self, images, *params)
        return self._encode(images)

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
