"""MAGVITv2 image tokenizer on the gfx950 fp32 kernels -- drop-in for the reference's
`models/multimodal_encoder/magvitv2.py` (`MAGVITv2.get_code / encode / decode_code`, same
state-dict keys: `encoder.*`, `decoder.*`, `quantize.{embedding,power_vals}`).

Architecture restated from the reference (magvitv2.py:57-178 encoder, :286-408 decoder,
common_modules.py ResnetBlock :301-360, AttnBlock :171-214, Downsample :76-93, Upsample :30-43,
Normalize :24-27): ch 128, encoder multipliers (1,2,2,4,4) with (4,3,4,3,4) res-blocks, decoder
multipliers (1,1,2,2,4) with (4,4,3,4,3), one single-head attention block in each `mid`, 13
latent channels, lookup-free quantiser with 2^13 codes.

Execution is NHWC fp32 end to end: every conv is an implicit GEMM on the f32 matrix cores
(`ug_conv2d_f32`), GroupNorm+swish is one stats pass + one apply pass, the residual add rides in
the conv epilogue, nearest-2x upsampling and the asymmetric stride-2 padding are folded into the
conv's gather.  torch modules below only HOLD parameters (names/shapes/initialisation); none of
their forward()s is ever called.
"""
import math
import os

import torch
import torch.nn as nn

from unigen_hip import ops
from unigen_hip.lib import UniGenHipError

from ..modeling_utils import ConfigMixin, ModelMixin, register_to_config

_CH = 128
_ENC_MULT, _ENC_BLOCKS = (1, 2, 2, 4, 4), (4, 3, 4, 3, 4)
_DEC_MULT, _DEC_BLOCKS = (1, 1, 2, 2, 4), (4, 4, 3, 4, 3)
_ZC = 13
_SPLIT_CONV = os.environ.get("UNIGEN_CONV_FP32_MFMA", "0") != "1"


def _gn(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class _Res(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = _gn(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = _gn(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1, 1, 0)


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = _gn(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)


class _Resample(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride, 0 if stride == 2 else 1)


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.block_1 = _Res(c, c)
        self.attn_1 = _Attn(c)
        self.block_2 = _Res(c, c)


class _Level(nn.Module):
    pass


class VQGANEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv_in = nn.Conv2d(3, _CH, 3, 1, 1)
        self.down = nn.ModuleList()
        cin = _CH
        for lvl, (mult, nblk) in enumerate(zip(_ENC_MULT, _ENC_BLOCKS)):
            level = _Level()
            level.block = nn.ModuleList()
            level.attn = nn.ModuleList()
            for _ in range(nblk):
                level.block.append(_Res(cin, _CH * mult))
                cin = _CH * mult
            if lvl != len(_ENC_MULT) - 1:
                level.downsample = _Resample(cin, 2)
            self.down.append(level)
        self.mid = _Mid(cin)
        self.norm_out = _gn(cin)
        self.conv_out = nn.Conv2d(cin, _ZC, 3, 1, 1)
        self.quant_conv = nn.Conv2d(_ZC, _ZC, 1)


class VQGANDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        cin = _CH * _DEC_MULT[-1]
        self.conv_in = nn.Conv2d(_ZC, cin, 3, 1, 1)
        self.mid = _Mid(cin)
        levels = {}
        for lvl in reversed(range(len(_DEC_MULT))):
            level = _Level()
            level.block = nn.ModuleList()
            level.attn = nn.ModuleList()
            for _ in range(_DEC_BLOCKS[lvl]):
                level.block.append(_Res(cin, _CH * _DEC_MULT[lvl]))
                cin = _CH * _DEC_MULT[lvl]
            if lvl != 0:
                level.upsample = _Resample(cin, 1)
            levels[lvl] = level
        self.up = nn.ModuleList([levels[i] for i in range(len(_DEC_MULT))])
        self.norm_out = _gn(cin)
        self.conv_out = nn.Conv2d(cin, 3, 3, 1, 1)
        self.post_quant_conv = nn.Conv2d(_ZC, _ZC, 1)


class LFQuantizer(nn.Module):
    def __init__(self, codebook_dim=_ZC):
        super().__init__()
        self.codebook_size = 2 ** codebook_dim
        self.e_dim = codebook_dim
        codes = torch.arange(self.codebook_size)
        bits = (codes.unsqueeze(1) >> torch.arange(codebook_dim - 1, -1, -1, dtype=torch.long)) & 1
        self.register_buffer("embedding", bits.float() * 2 - 1)
        self.register_buffer("power_vals", 2 ** torch.arange(codebook_dim - 1, -1, -1))


class _Packed:
    """Per-conv packed weights [k*k][Cin][cout_pad] + bias, built once per weight version."""
    __slots__ = ("w", "cpad", "bias", "cout", "cin", "k", "ws")


class MAGVITv2(ModelMixin, ConfigMixin):
    @register_to_config
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.encoder = VQGANEncoder()
        self.decoder = VQGANDecoder()
        self.quantize = LFQuantizer()
        self._packed = {}
        self._err = None
        # True: every conv on the exact fp32 MFMA chain (ug_conv2d_f32) and stand-alone GroupNorm passes;
        # False (default): wide convs on the split-f16 contraction.  Flip it on an instance to compare the two.
        self.exact_fp32_convs = not _SPLIT_CONV
        # GroupNorm sums gathered in the producing convolution's epilogue (UNIGEN_GN_FUSE_STATS=0: a separate pass per norm)
        self.fuse_gn_stats = os.environ.get("UNIGEN_GN_FUSE_STATS", "1") != "0"

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, *a, **k):
        self.__dict__["_packed"] = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.__dict__["_packed"] = {}
        return super().load_state_dict(*a, **k)

    def _pk(self, conv, pad_cin_to=None):
        key = (id(conv), self.exact_fp32_convs, pad_cin_to)
        ver = conv.weight._version
        hit = self._packed.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        w = conv.weight.detach().float()
        if not w.is_cuda:
            raise UniGenHipError("MAGVITv2 must live in GPU memory (.to('cuda')); there is no CPU implementation")
        if pad_cin_to is not None and w.shape[1] < pad_cin_to:
            w = torch.cat([w, w.new_zeros(w.shape[0], pad_cin_to - w.shape[1], *w.shape[2:])], 1)
        p = _Packed()
        p.w, p.cpad = ops.pack_conv_weight(w)
        p.bias = conv.bias.detach().float().contiguous()
        p.cout, p.cin, p.k = w.shape[0], w.shape[1], w.shape[2]
        # wide convs run on the f16 matrix cores with scaled two-way split operands (fp32-accurate, 16/3 the MFMA rate);
        # UNIGEN_CONV_FP32_MFMA=1 keeps every conv on the exact fp32 MFMA chain
        p.ws = ops.split_conv_weight(p.w) if (not self.exact_fp32_convs and ops.conv_split_eligible(p.cin, p.cout, p.cpad)) else None
        self._packed[key] = (ver, p)
        return p

    def _conv(self, x, conv, residual=None, upsample=False, asym=False, pad_cin_to=None, norm=None, swish=True, stats=False):
        """conv(x), or conv(swish?(norm(x))) when `norm` (a GroupNorm) is given.  Wide 3x3 stride-1 convs take the
        LDS-resident-patch kernel, which applies the normalisation on its load path and -- `stats=True`: the output feeds a
        GroupNorm(32) -- gathers that norm's sums while it stores (the result carries them as `_ug_gn_stats`)."""
        p = self._pk(conv, pad_cin_to)
        if p.ws is not None and p.k == 3 and not upsample and not asym:
            gn, bound = None, None
            st_in = getattr(x, "_ug_gn_stats", None)
            if norm is not None:
                mr = (ops.groupnorm_finalize(st_in, x.shape[0], x.shape[1] * x.shape[2], x.shape[3], groups=32, eps=norm.eps)
                      if st_in is not None else ops.groupnorm_stats(x, groups=32, eps=norm.eps))
                gn = (mr, norm.weight.detach(), norm.bias.detach(), 32, swish)
                bound = self._gn_bound(norm, (x.shape[3] // 32) * x.shape[1] * x.shape[2])
            st = self._stats_slot(x) if stats and p.cout % 128 == 0 else None
            y = ops.conv3x3_nhwc(x, p.ws, p.cpad, p.bias, p.cout, residual=residual, gn=gn, gn_bound=bound, out_stats=st,
                                 x_amax=ops.stats_amax(st_in) if norm is None and st_in is not None else None)
            if st is not None:
                y._ug_gn_stats = st
            return y
        if norm is not None:
            x = self._norm(x, norm, swish)
        st_in = getattr(x, "_ug_gn_stats", None) if p.ws is not None else None       # (a normalised x is a new tensor: no attribute)
        st = None
        if stats and p.ws is not None and p.cout % 128 == 0:
            Ho, Wo = (x.shape[1] // 2, x.shape[2] // 2) if asym else (x.shape[1] * 2, x.shape[2] * 2) if upsample else x.shape[1:3]
            st = self._stats_slot(x) if (Ho * Wo) % 128 == 0 else None
        y = ops.conv2d_nhwc(x, p.w, p.cpad, p.bias, p.cout, p.k, stride=2 if asym else 1, asym_pad=asym, upsample=upsample,
                            residual=residual, w_split=p.ws, out_stats=st, x_amax=ops.stats_amax(st_in) if st_in is not None else None)
        if st is not None:
            y._ug_gn_stats = st
        return y

    def _stats_slot(self, x):
        """a zeroed statistics buffer for a convolution whose output feeds a GroupNorm(32) (one pooled allocation per pass)"""
        if not self.fuse_gn_stats:
            return None
        pool = self.__dict__.get("_stats_pool")
        if not pool or pool[0] != x.shape[0] or not pool[1]:
            pool = (x.shape[0], ops.gn_stats_slots(64, x.shape[0], x.device))
            self.__dict__["_stats_pool"] = pool
        return pool[1].pop()

    def _gn_bound(self, norm, group_elems):
        """scale bound of swish?(norm(x)) for the split convolution that applies `norm` on its load path, from the layer's own
        gamma / beta (two device reads per weight version, cached like the packed conv weights)"""
        key = ("gn", id(norm))
        ver = (norm.weight._version, norm.bias._version)
        hit = self._packed.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, float(norm.weight.detach().abs().max()), float(norm.bias.detach().abs().max()))
            self._packed[key] = hit
        return ops.gn_out_bound(hit[1], hit[2], group_elems)

    @staticmethod
    def _norm(x, gn, swish=True):
        return ops.groupnorm_swish(x, gn.weight.detach(), gn.bias.detach(), groups=32, eps=gn.eps, swish=swish)

    def _res(self, x, blk):
        h = self._conv(x, blk.conv1, norm=blk.norm1, stats=True)                 # read by norm2
        skip = self._conv(x, blk.nin_shortcut) if hasattr(blk, "nin_shortcut") else x
        return self._conv(h, blk.conv2, residual=skip, norm=blk.norm2, stats=True)   # read by the next block's norm1 / norm_out

    def _attn(self, x, a):
        B, H, W, C = x.shape
        T = H * W
        hn = self._norm(x, a.norm, swish=False)
        q, k, v = self._conv(hn, a.q), self._conv(hn, a.k), self._conv(hn, a.v)
        s = ops.gemm_f32(q, k, b_is_nk=True, M=T, N=T, K=C, batch=B, lda=C, ldb=C, stride_a=T * C, stride_b=T * C)
        ops.softmax_rows_(s.view(B * T, T), float(int(C) ** (-0.5)))
        ctx = ops.gemm_f32(s, v, b_is_nk=False, M=T, N=C, K=T, batch=B, lda=T, ldb=C, stride_a=T * T, stride_b=T * C)
        return self._conv(ctx.view(B, H, W, C), a.proj_out, residual=x)

    def _mid(self, x, mid):
        return self._res(self._attn(self._res(x, mid.block_1), mid.attn_1), mid.block_2)

    # ------------------------------------------------------------------ encoder / decoder graphs
    @torch.no_grad()
    def _encode_z(self, pixel_values):
        """NCHW fp32 image -> pre-quantisation latents, NHWC [B, 16, 16, 13]."""
        e = self.encoder
        # RGB padded with zero channels: to one 32-channel slab of the LDS-resident-patch kernel (a quarter of a 128 -> 128 layer's
        # time, and the first GroupNorm's sums come with it; the exact fp32 im2col kernel on 4-channel pixels wrote its 537 MB
        # output at 0.9 TB/s: 0.62 ms + a 0.15 ms statistics pass), or to 4 channels (16-byte pixels) on the exact-fp32 path
        x = ops.nchw_to_nhwc(pixel_values.float().contiguous(), 4)         # 16-byte pixels; the kernel reads the other 28 as zero
        h = self._conv(x, e.conv_in, pad_cin_to=4 if self.exact_fp32_convs or not self.fuse_gn_stats else 32, stats=True)
        for lvl, level in enumerate(e.down):
            for blk in level.block:
                h = self._res(h, blk)
            if hasattr(level, "downsample"):
                h = self._conv(h, level.downsample.conv, asym=True, stats=True)
        h = self._mid(h, e.mid)
        h = self._conv(h, e.conv_out, norm=e.norm_out)
        return self._conv(h, e.quant_conv)

    @torch.no_grad()
    def _decode_z(self, z_nhwc):
        d = self.decoder
        h = self._conv(self._conv(z_nhwc, d.post_quant_conv), d.conv_in)
        h = self._mid(h, d.mid)
        for lvl in reversed(range(len(d.up))):
            level = d.up[lvl]
            for blk in level.block:
                h = self._res(h, blk)
            if hasattr(level, "upsample"):
                h = self._conv(h, level.upsample.conv, upsample=True, stats=True)
        return self._conv(h, d.conv_out, norm=d.norm_out)

    # ------------------------------------------------------------------ reference API
    def forward(self, pixel_values, return_loss=False):
        pass

    @torch.no_grad()
    def encode(self, pixel_values, return_loss=False):
        z = self._encode_z(pixel_values)
        B = z.shape[0]
        idx = ops.lfq_pack(z.view(-1, _ZC), _ZC).view(B, -1)
        zq = ops.lfq_unpack(idx.view(-1), _ZC).view(B, z.shape[1], z.shape[2], _ZC).permute(0, 3, 1, 2).contiguous()
        return zq, idx

    @torch.no_grad()
    def get_code(self, pixel_values):
        z = self._encode_z(pixel_values)
        return ops.lfq_pack(z.view(-1, _ZC), _ZC).view(z.shape[0], -1)

    @torch.no_grad()
    def get_latents(self, pixel_values):
        """Test hook: pre-quantisation z in the reference's NCHW layout [B, 13, 16, 16]."""
        return self._encode_z(pixel_values).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def decode_code(self, codebook_indices, shape=None):
        b, n = codebook_indices.shape
        h, w = (int(math.sqrt(n)), int(math.sqrt(n))) if shape is None else shape
        if self._err is None or self._err.device != codebook_indices.device:
            self._err = torch.zeros(1, dtype=torch.int32, device=codebook_indices.device)
        z = ops.lfq_unpack(codebook_indices.reshape(-1).long().contiguous(), _ZC, self._err).view(b, h, w, _ZC)
        out = self._decode_z(z)
        return ops.nhwc_to_nchw(out, 3)
