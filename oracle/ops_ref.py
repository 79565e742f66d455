"""ORACLE (test infrastructure, not product code): plain-torch CPU restatements of the individual
operators on the UniGen hot path, each citing the reference / third-party lines it follows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The shipped path (ml-unigen_amd/) never does; it fails loudly when its HIP extension is missing.

Pinned against the real reference by tools/make_golden.py (run in the build container, where
/root/reference is importable) -> tests/golden/*.pt, checked by tests/test_oracle_golden.py.
"""
import math
import torch
import torch.nn.functional as F


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


# transformers Qwen2RMSNorm.forward (modeling_qwen2.py:246-252): fp32 statistics, weight applied last.
def rmsnorm_ref(x, w, eps):
    x = x.to(torch.float32)
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rmsnorm_bwd_ref(dy, x, w, eps):
    """returns (dx, dw) of y = rmsnorm(x) * w for upstream dy (all fp32)."""
    x = x.detach().clone().requires_grad_(True)
    w = w.detach().clone().requires_grad_(True)
    y = rmsnorm_ref(x, w, eps)
    y.backward(dy)
    return x.grad, w.grad


# Qwen2RotaryEmbedding.forward (modeling_qwen2.py:91-102) for position_ids = arange(L)
def rope_tables_ref(L, head_dim, theta):
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    pos = torch.arange(L, dtype=torch.float)
    freqs = (inv_freq[None, :, None] @ pos[None, None, :]).transpose(1, 2)   # [1, L, d/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos()[0], emb.sin()[0]          # [L, d]


def rotate_half_ref(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


# apply_rotary_pos_emb (modeling_qwen2.py:113-135); x [B, heads, L, d], cos/sin [L, d] fp32
def rope_ref(x, cos, sin):
    return (x * cos) + (rotate_half_ref(x) * sin)


# Qwen2MLP.forward inner product (modeling_qwen2.py:46-48) under bf16 autocast: silu and the
# product are separate bf16 ops.  gu = [gate | up] bf16.
def swiglu_ref(gu):
    i = gu.shape[-1] // 2
    g, u = gu[..., :i], gu[..., i:]
    return F.silu(g) * u


# eager_attention_forward (modeling_qwen2.py:152-173) with GQA repeat_kv; q [B,H,L,d] k,v [B,HKV,L,d]
# mask_add [B,1,L,L] additive.  All math fp32 here (the tolerance of the bf16 kernel is judged
# against this).
def attention_ref(q, k, v, mask_add, scale):
    B, H, L, d = q.shape
    rep = H // k.shape[1]
    k = k.repeat_interleave(rep, dim=1)
    v = v.repeat_interleave(rep, dim=1)
    w = torch.matmul(q.float(), k.float().transpose(2, 3)) * scale
    if mask_add is not None:
        w = w + mask_add.float()
    w = torch.softmax(w, dim=-1, dtype=torch.float32)
    return torch.matmul(w, v.float())


# F.cross_entropy(..., ignore_index=-100) as called at models/unigen.py:310-338
def ce_ref(logits, labels, ignore_index=-100):
    return F.cross_entropy(logits.float(), labels, ignore_index=ignore_index)


# LFQuantizer.get_indices (magvitv2.py:210-215): channel 0 is the MSB; z == 0 counts as bit 0
def lfq_indices_ref(z_nchw):
    nb = z_nchw.shape[1]
    pw = 2 ** torch.arange(nb - 1, -1, -1)
    return (pw.reshape(1, -1, 1, 1) * (z_nchw > 0).float()).sum(1).long().reshape(z_nchw.shape[0], -1)


# LFQuantizer.get_codebook_entry (magvitv2.py:217-230)
def lfq_entries_ref(idx, nbits):
    b, n = idx.shape
    h = w = int(math.sqrt(n))
    bits = ((idx.reshape(-1, 1) >> torch.arange(nbits - 1, -1, -1)) & 1).float() * 2 - 1
    return bits.view(b, h, w, nbits).permute(0, 3, 1, 2).contiguous()
