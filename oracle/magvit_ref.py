"""ORACLE (test infrastructure): plain-torch CPU restatement of the MAGVITv2 tokenizer, written as
functions over the reference checkpoint's state dict (keys `encoder.*`, `decoder.*`).

Follows models/multimodal_encoder/magvitv2.py: VQGANEncoder.forward :152-178, LFQuantizer
:210-230, VQGANDecoder.forward :374-408, MAGVITv2.get_code/decode_code :432-442, and
common_modules.py: nonlinearity :19-21, Normalize :24-27, Upsample :39-43, Downsample :86-93,
AttnBlock.forward :190-214, ResnetBlock.forward :340-360.  fp32 throughout, like the reference
(the tokenizer is never cast and runs outside autocast).
"""
import math

import torch
import torch.nn.functional as F

ENC_BLOCKS = (4, 3, 4, 3, 4)
DEC_BLOCKS = (4, 4, 3, 4, 3)
ZC = 13


def _conv(sd, p, x, stride=1, padding=None):
    w = sd[p + ".weight"]
    pad = (w.shape[-1] // 2) if padding is None else padding
    return F.conv2d(x, w, sd[p + ".bias"], stride=stride, padding=pad)


def _norm(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _res(sd, p, x):
    h = _conv(sd, p + ".conv1", _swish(_norm(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", _swish(_norm(sd, p + ".norm2", h)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def _attn(sd, p, x):
    h = _norm(sd, p + ".norm", x)
    q, k, v = _conv(sd, p + ".q", h), _conv(sd, p + ".k", h), _conv(sd, p + ".v", h)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h)


def _mid(sd, p, x):
    return _res(sd, p + ".block_2", _attn(sd, p + ".attn_1", _res(sd, p + ".block_1", x)))


def encode_z_ref(sd, pixels):
    """[B,3,256,256] fp32 -> pre-quantisation latents [B,13,16,16]."""
    h = _conv(sd, "encoder.conv_in", pixels)
    for lvl, nblk in enumerate(ENC_BLOCKS):
        for j in range(nblk):
            h = _res(sd, f"encoder.down.{lvl}.block.{j}", h)
        if lvl != len(ENC_BLOCKS) - 1:
            h = _conv(sd, f"encoder.down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = _mid(sd, "encoder.mid", h)
    h = _conv(sd, "encoder.conv_out", _swish(_norm(sd, "encoder.norm_out", h)))
    return _conv(sd, "encoder.quant_conv", h)


def get_code_ref(sd, pixels):
    z = encode_z_ref(sd, pixels)
    pw = 2 ** torch.arange(ZC - 1, -1, -1)
    return (pw.reshape(1, -1, 1, 1) * (z > 0).float()).sum(1).long().reshape(pixels.shape[0], -1)


def decode_code_ref(sd, idx):
    b, n = idx.shape
    s = int(math.sqrt(n))
    bits = ((idx.reshape(-1, 1) >> torch.arange(ZC - 1, -1, -1)) & 1).float() * 2 - 1
    z = bits.view(b, s, s, ZC).permute(0, 3, 1, 2).contiguous()
    h = _conv(sd, "decoder.conv_in", _conv(sd, "decoder.post_quant_conv", z))
    h = _mid(sd, "decoder.mid", h)
    for lvl in reversed(range(len(DEC_BLOCKS))):
        for j in range(DEC_BLOCKS[lvl]):
            h = _res(sd, f"decoder.up.{lvl}.block.{j}", h)
        if lvl != 0:
            h = _conv(sd, f"decoder.up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(sd, "decoder.conv_out", _swish(_norm(sd, "decoder.norm_out", h)))


def magvit_param_shapes():
    """(name, shape) of every parameter of the reference MAGVITv2, derived from the architecture
    constants (ch 128; encoder mult (1,2,2,4,4); decoder mult (1,1,2,2,4))."""
    out = []

    def conv(p, cin, cout, k):
        out.append((p + ".weight", (cout, cin, k, k)))
        out.append((p + ".bias", (cout,)))

    def gn(p, c):
        out.append((p + ".weight", (c,)))
        out.append((p + ".bias", (c,)))

    def res(p, cin, cout):
        gn(p + ".norm1", cin); conv(p + ".conv1", cin, cout, 3)
        gn(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cin, cout, 1)

    def mid(p, c):
        res(p + ".block_1", c, c)
        gn(p + ".attn_1.norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(p + ".attn_1." + n, c, c, 1)
        res(p + ".block_2", c, c)

    ch = 128
    conv("encoder.conv_in", 3, ch, 3)
    cin = ch
    for lvl, (m, nb) in enumerate(zip((1, 2, 2, 4, 4), ENC_BLOCKS)):
        for j in range(nb):
            res(f"encoder.down.{lvl}.block.{j}", cin, ch * m)
            cin = ch * m
        if lvl != 4:
            conv(f"encoder.down.{lvl}.downsample.conv", cin, cin, 3)
    mid("encoder.mid", cin)
    gn("encoder.norm_out", cin)
    conv("encoder.conv_out", cin, ZC, 3)
    conv("encoder.quant_conv", ZC, ZC, 1)
    dm = (1, 1, 2, 2, 4)
    cin = ch * dm[-1]
    conv("decoder.conv_in", ZC, cin, 3)
    mid("decoder.mid", cin)
    for lvl in reversed(range(5)):
        for j in range(DEC_BLOCKS[lvl]):
            res(f"decoder.up.{lvl}.block.{j}", cin, ch * dm[lvl])
            cin = ch * dm[lvl]
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", cin, cin, 3)
    gn("decoder.norm_out", cin)
    conv("decoder.conv_out", cin, 3, 3)
    conv("decoder.post_quant_conv", ZC, ZC, 1)
    return out
