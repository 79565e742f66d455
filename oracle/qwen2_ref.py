"""ORACLE (test infrastructure): plain-torch CPU restatement of the Qwen2.5 backbone + UniGen's
multi-task forward, i.e. what `UniGen.forward` executes through transformers' Qwen2ForCausalLM.

Follows, line by line:
  * models/unigen.py:240-342 (embed / inputs_embeds, backbone, tied lm_head on ALL positions, the
    three masked cross-entropies and their slicing),
  * transformers modeling_qwen2.py: Qwen2RMSNorm :238-252, Qwen2RotaryEmbedding :51-102 (default
    rope, position_ids = arange(L)), apply_rotary_pos_emb :113-135, Qwen2Attention :176-234 with the
    `sdpa` backend (integrations/sdpa_attention.py: repeat_kv + F.scaled_dot_product_attention with the
    caller's 4-D additive mask, is_causal=False), Qwen2MLP :35-48, Qwen2DecoderLayer :258-298,
    Qwen2Model.forward :331-400.
Third-party pin: the reference pins transformers==4.51.0; this restatement was validated against
the 5.15.0 present in the build container (same arithmetic for this path).

Parameter names equal the reference checkpoint's `llm.*` keys so the reference's state_dict loads
unchanged.  Precision mode A (accelerate DDP + bf16 autocast) is reproduced by running the same
code under torch.autocast('cpu', dtype=torch.bfloat16) exactly like accelerate wraps the reference.
"""
import contextlib
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops_ref import rope_ref


class Qwen2Cfg:
    def __init__(self, vocab_size, hidden_size=1536, intermediate_size=8960, num_hidden_layers=28,
                 num_attention_heads=12, num_key_value_heads=2, head_dim=None, rope_theta=1e6, rms_norm_eps=1e-6,
                 initializer_range=0.02, rope_scaling=None, max_position_embeddings=32768):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.head_dim = head_dim or hidden_size // num_attention_heads
        self.rope_theta = rope_theta
        self.rms_norm_eps = rms_norm_eps
        self.initializer_range = initializer_range
        self.rope_scaling = rope_scaling            # {"factor": f, "type": "linear" | "dynamic"} (models/unigen.py:63-64) or None
        self.max_position_embeddings = max_position_embeddings

    def to_hf_dict(self):
        return dict(architectures=["Qwen2ForCausalLM"], model_type="qwen2", vocab_size=self.vocab_size,
                    hidden_size=self.hidden_size, intermediate_size=self.intermediate_size,
                    num_hidden_layers=self.num_hidden_layers, num_attention_heads=self.num_attention_heads,
                    num_key_value_heads=self.num_key_value_heads, rope_theta=self.rope_theta,
                    rms_norm_eps=self.rms_norm_eps, hidden_act="silu", max_position_embeddings=32768,
                    tie_word_embeddings=True, attention_dropout=0.0, use_sliding_window=False,
                    initializer_range=self.initializer_range, torch_dtype="float32", use_cache=False)


QWEN25_1P5B = dict(hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12,
                   num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6)


class _Norm(nn.Module):
    def __init__(self, n, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.eps = eps

    def forward(self, x):
        dt = x.dtype
        xf = x.to(torch.float32)
        xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)
        return self.weight * xf.to(dt)


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        hd = c.head_dim
        self.q_proj = nn.Linear(c.hidden_size, c.num_attention_heads * hd, bias=True)
        self.k_proj = nn.Linear(c.hidden_size, c.num_key_value_heads * hd, bias=True)
        self.v_proj = nn.Linear(c.hidden_size, c.num_key_value_heads * hd, bias=True)
        self.o_proj = nn.Linear(c.num_attention_heads * hd, c.hidden_size, bias=False)

    def forward(self, x, cos, sin, mask, cache=None):
        c = self.c
        B, L, _ = x.shape
        hd = c.head_dim
        q = self.q_proj(x).view(B, L, -1, hd).transpose(1, 2)
        k = self.k_proj(x).view(B, L, -1, hd).transpose(1, 2)
        v = self.v_proj(x).view(B, L, -1, hd).transpose(1, 2)
        q, k = rope_ref(q, cos, sin), rope_ref(k, cos, sin)
        if cache is not None:                         # DynamicCache: concatenate along the sequence
            if cache.get("k") is not None:
                k = torch.cat([cache["k"], k], dim=2)
                v = torch.cat([cache["v"], v], dim=2)
            cache["k"], cache["v"] = k, v
        rep = c.num_attention_heads // c.num_key_value_heads
        k = k[:, :, None].expand(B, k.shape[1], rep, k.shape[2], hd).reshape(B, -1, k.shape[2], hd)
        v = v[:, :, None].expand(B, v.shape[1], rep, v.shape[2], hd).reshape(B, -1, v.shape[2], hd)
        causal = mask is None and L > 1
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, scale=hd ** -0.5, is_causal=causal)
        return self.o_proj(o.transpose(1, 2).reshape(B, L, -1).contiguous())


class _MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gate_proj = nn.Linear(c.hidden_size, c.intermediate_size, bias=False)
        self.up_proj = nn.Linear(c.hidden_size, c.intermediate_size, bias=False)
        self.down_proj = nn.Linear(c.intermediate_size, c.hidden_size, bias=False)

    def forward(self, x):
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self_attn = _Attn(c)
        self.mlp = _MLP(c)
        self.input_layernorm = _Norm(c.hidden_size, c.rms_norm_eps)
        self.post_attention_layernorm = _Norm(c.hidden_size, c.rms_norm_eps)

    def forward(self, h, cos, sin, mask, cache=None):
        h = h + self.self_attn(self.input_layernorm(h), cos, sin, mask, cache)
        return h + self.mlp(self.post_attention_layernorm(h))


class _Backbone(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embed_tokens = nn.Embedding(c.vocab_size, c.hidden_size)
        self.layers = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])
        self.norm = _Norm(c.hidden_size, c.rms_norm_eps)


class RefCausalLM(nn.Module):
    """state_dict keys: model.embed_tokens.weight, model.layers.N.*, model.norm.weight, lm_head.weight (tied)."""

    def __init__(self, c):
        super().__init__()
        self.cfg = c
        self.model = _Backbone(c)
        self.lm_head = nn.Linear(c.hidden_size, c.vocab_size, bias=False)
        self.lm_head.weight = self.model.embed_tokens.weight

    def rope(self, L, dtype, offset=0):
        c = self.cfg
        base, sc = c.rope_theta, (c.rope_scaling or {})
        kind, factor = sc.get("rope_type", sc.get("type", "default")), float(sc.get("factor", 1.0))
        if kind == "dynamic":           # modeling_rope_utils._compute_dynamic_ntk_parameters; the sequence length arrives as an
            # int64 TENSOR (max(position_ids) + 1), so the new base is fp32 tensor arithmetic, not Python doubles
            sl = torch.maximum(torch.tensor(offset + L), torch.tensor(c.max_position_embeddings))
            base = base * ((factor * sl / c.max_position_embeddings) - (factor - 1)) ** (c.head_dim / (c.head_dim - 2))
        inv_freq = 1.0 / (base ** (torch.arange(0, c.head_dim, 2, dtype=torch.float) / c.head_dim))
        if kind == "linear":            # _compute_linear_scaling_rope_parameters
            inv_freq = inv_freq / factor
        pos = torch.arange(offset, offset + L, dtype=torch.float)
        with torch.autocast("cpu", enabled=False):
            freqs = (inv_freq[None, :, None].float() @ pos[None, None, :].float()).transpose(1, 2)
            emb = torch.cat((freqs, freqs), dim=-1)
            cos, sin = emb.cos(), emb.sin()
        return cos.to(dtype)[:, None], sin.to(dtype)[:, None]      # [1,1,L,d] broadcast over heads

    def backbone(self, input_ids=None, inputs_embeds=None, mask=None, caches=None, pos_offset=0):
        h = self.model.embed_tokens(input_ids) if inputs_embeds is None else inputs_embeds
        cos, sin = self.rope(h.shape[1], h.dtype, pos_offset)
        for i, layer in enumerate(self.model.layers):
            h = layer(h, cos, sin, mask, None if caches is None else caches[i])
        return self.model.norm(h)

    def forward(self, input_ids=None, inputs_embeds=None, mask=None):
        return self.lm_head(self.backbone(input_ids, inputs_embeds, mask))


def init_like_hf(model, seed):
    """HF Qwen2 `_init_weights`: N(0, initializer_range) for Linear/Embedding weights, zeros for
    biases, ones for norms (what Qwen2ForCausalLM(config) does at models/unigen.py:65)."""
    g = torch.Generator().manual_seed(seed)
    std = model.cfg.initializer_range
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
                p.fill_(1.0)
            elif name.endswith(".bias"):
                p.zero_()
            else:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model


def autocast_ctx(enabled):
    return torch.autocast("cpu", dtype=torch.bfloat16) if enabled else contextlib.nullcontext()


def unigen_forward_ref(lm, input_ids, attention_mask, labels=None, input_embeddings=None, batch_size_t2i=0,
                       batch_size_lm=0, batch_size_mmu=0, num_vq_tokens=256, t2i_mode="mask", autocast=True):
    """UniGen.forward (models/unigen.py:240-342), gen_proj_depth == 0 branch (the only one any shipped
    config reaches).  Returns logits if labels is None else (logits, loss_t2i, loss_lm, loss_mmu);
    under autocast the returned tensors are converted to fp32 like accelerate's forward wrapper does."""
    V = lm.cfg.vocab_size
    with autocast_ctx(autocast):
        logits = lm(input_ids if input_embeddings is None else None, input_embeddings, attention_mask)
        if labels is None:
            return logits.float() if autocast else logits
        n = num_vq_tokens
        if t2i_mode == "mask":
            loss_t2i = F.cross_entropy(logits[:batch_size_t2i, -(n + 1):-1].contiguous().view(-1, V),
                                       labels[:batch_size_t2i, -(n + 1):-1].contiguous().view(-1), ignore_index=-100)
        else:
            loss_t2i = F.cross_entropy(logits[:batch_size_t2i, -(n + 2):-1].contiguous().view(-1, V),
                                       labels[:batch_size_t2i, -(n + 1):].contiguous().view(-1), ignore_index=-100)
        loss_lm = 0.0
        if batch_size_lm > 0:
            loss_lm = F.cross_entropy(logits[batch_size_t2i:batch_size_t2i + batch_size_lm, :-1].contiguous().view(-1, V),
                                      labels[batch_size_t2i:batch_size_t2i + batch_size_lm, 1:].contiguous().view(-1),
                                      ignore_index=-100)
        loss_mmu = 0.0
        if batch_size_mmu > 0:
            loss_mmu = F.cross_entropy(logits[-batch_size_mmu:, :-1].contiguous().view(-1, V),
                                       labels[-batch_size_mmu:, 1:].contiguous().view(-1), ignore_index=-100)
    if autocast:
        logits = logits.float()
        loss_t2i = loss_t2i.float()
        loss_lm = loss_lm.float() if torch.is_tensor(loss_lm) else loss_lm
        loss_mmu = loss_mmu.float() if torch.is_tensor(loss_mmu) else loss_mmu
    return logits, loss_t2i, loss_lm, loss_mmu


class GenHeadRef(nn.Module):
    """The gen_projector path's extra modules (models/unigen.py:74-90): gen_embed (codebook + 1 rows), gen_projector
    (Linear -> GELU -> Linear ...), img_head (hidden -> codebook, no bias).  State-dict keys as in the reference."""

    def __init__(self, hidden, codebook, depth=2, use_gen_dim=False, gen_input_dim=16):
        super().__init__()
        if use_gen_dim:
            self.gen_embed = nn.Embedding(codebook + 1, gen_input_dim)
            mods, width = [nn.Linear(gen_input_dim, hidden)], hidden
        else:
            self.gen_embed = nn.Embedding(codebook + 1, hidden)
            mods, width = [nn.Linear(hidden, hidden * 2)], hidden * 2
        for _ in range(1, depth):
            mods += [nn.GELU(), nn.Linear(width, hidden)]
            width = hidden
        self.gen_projector = nn.Sequential(*mods)
        self.img_head = nn.Linear(hidden, codebook, bias=False)


def unigen_forward_gen_ref(lm, gen, input_ids, attention_mask, labels=None, batch_size_t2i=0, batch_size_lm=0,
                           batch_size_mmu=0, num_vq_tokens=256, t2i_mode="mask", autocast=True):
    """UniGen.forward, gen_proj_depth > 0 branch (models/unigen.py:255-270, 301-341): image slots embedded by
    gen_projector(gen_embed(raw codes)), img_head on the t2i rows, tied lm_head on the remaining rows."""
    n, bt = num_vq_tokens, batch_size_t2i
    V, C = lm.cfg.vocab_size, gen.img_head.out_features
    with autocast_ctx(autocast):
        emb = lm.model.embed_tokens(input_ids)
        img = gen.gen_projector(gen.gen_embed(input_ids[:, -(n + 1):-1].contiguous()))
        emb = torch.cat([emb[:, :-(n + 1)], img.to(emb.dtype), emb[:, -1:]], 1)        # = the in-place slot assignment
        hidden = lm.backbone(None, emb, attention_mask)
        img_logits = gen.img_head(hidden[:bt])
        if labels is None:
            return img_logits.float() if autocast else img_logits
        logits = lm.lm_head(hidden[bt:])
        if t2i_mode == "mask":
            loss_t2i = F.cross_entropy(img_logits[:, -(n + 1):-1].contiguous().view(-1, C),
                                       labels[:bt, -(n + 1):-1].contiguous().view(-1), ignore_index=-100)
        else:
            loss_t2i = F.cross_entropy(img_logits[:, -(n + 2):-1].contiguous().view(-1, C),
                                       labels[:bt, -(n + 1):].contiguous().view(-1), ignore_index=-100)
        loss_lm = 0.0
        if batch_size_lm > 0:
            loss_lm = F.cross_entropy(logits[:batch_size_lm, :-1].contiguous().view(-1, V),
                                      labels[bt:bt + batch_size_lm, 1:].contiguous().view(-1), ignore_index=-100)
        loss_mmu = 0.0
        if batch_size_mmu > 0:
            loss_mmu = F.cross_entropy(logits[-batch_size_mmu:, :-1].contiguous().view(-1, V),
                                       labels[-batch_size_mmu:, 1:].contiguous().view(-1), ignore_index=-100)
    f = (lambda t: t.float() if torch.is_tensor(t) else t) if autocast else (lambda t: t)
    return f(img_logits), f(loss_t2i), f(loss_lm), f(loss_mmu)


def ar_generate_ref(lm, cond_embeds, uncond_embeds, n_tokens, guidance_scale, text_vocab, key_valid=None, autocast=True, gen=None,
                    trace=None, force_tokens=None):
    """Greedy (argmax) version of UniGen.t2i_generate_ar (models/unigen.py:457-521): prefix = embeddings with the
    last n+1 positions already cut off; KV cache grown by concatenation like DynamicCache; CFG
    `uncond + s * (cond - uncond)` on logits[text_vocab:-1]; next token embedded for both halves.
    gen (a GenHeadRef): the gen_proj_depth > 0 branch (:486-495,512-514) -- img_head on the last hidden state (codebook-wide
    logits, no slicing, no text-vocabulary offset), next input gen_projector(gen_embed(token)).
    trace (a list): receives per step a dict with the head's code-book logits of all 2 B rows (`logits`, fp32 copy of what the head
    returned) and the CFG mix (`mixed`).  force_tokens [B, n]: the trajectory to follow instead of this run's own argmax (teacher
    forcing: lets an fp32 run be compared step by step with a bf16 run's trajectory); the returned tokens are still this run's argmax.
    Returns (tokens [B, n], top-2 margin of the mixed logits per step [B, n])."""
    B = cond_embeds.shape[0]
    x = torch.cat([cond_embeds, uncond_embeds])
    caches = [dict() for _ in lm.model.layers]
    toks, margins = [], []
    pos = 0
    with torch.no_grad(), autocast_ctx(autocast):
        for i in range(n_tokens):
            L_new = x.shape[1]
            total = pos + L_new
            if i == 0:
                r = torch.arange(total)
                allow = (r[None, :] <= r[:, None])[None, None].expand(2 * B, 1, total, total).clone()
                if key_valid is not None:
                    allow = allow & key_valid[:, None, None, :total].bool()
                mask = torch.where(allow, 0.0, float("-inf"))
                mask = torch.where(allow.any(-1, keepdim=True), mask, torch.zeros(()))    # fully padded rows: harmless
            else:
                allow = torch.ones(2 * B, 1, 1, total, dtype=torch.bool)
                if key_valid is not None:
                    kv = torch.ones(2 * B, total, dtype=torch.bool)
                    kv[:, :key_valid.shape[1]] = key_valid.bool()
                    allow = allow & kv[:, None, None, :]
                mask = torch.where(allow, 0.0, float("-inf"))
            h = lm.backbone(inputs_embeds=x, mask=mask, caches=caches, pos_offset=pos)
            # (gen branch: the img_head output stays in the autocast dtype, the CFG mix below is bf16 arithmetic there, :498-500)
            logits = gen.img_head(h[:, -1]) if gen is not None else lm.lm_head(h[:, -1]).float()[:, text_vocab:-1]
            cond, uncond = logits[:B], logits[B:]
            mixed = (uncond + guidance_scale * (cond - uncond)).float()
            top2 = mixed.topk(2, -1).values
            nxt = mixed.argmax(-1, keepdim=True)
            toks.append(nxt)
            margins.append(top2[:, 0] - top2[:, 1])
            if trace is not None:
                trace.append({"logits": logits.float().clone(), "mixed": mixed.clone()})
            if force_tokens is not None:
                nxt = force_tokens[:, i:i + 1].to(nxt.dtype)
            pos = total
            x = gen.gen_projector(gen.gen_embed(torch.cat([nxt, nxt]))) if gen is not None else lm.model.embed_tokens(torch.cat([nxt, nxt]) + text_vocab)
    return torch.cat(toks, 1), torch.stack(margins, 1)


def mmu_generate_ref(lm, idx=None, input_embeddings=None, attention_mask=None, max_new_tokens=100, eot_token=None,
                     autocast=True):
    """Greedy (temperature == 0) UniGen.mmu_generate (models/unigen.py:523-581), statement by statement: every new token
    re-runs the WHOLE sequence (no KV cache); the additive [1,1,L,L] mask grows by one column of
    finfo(logits.dtype).min and one row that copies the previous last row and ends in 0 (:543-558); the next token is
    argmax(logits[:, -1]) (:570); it is appended as an id (w_und_encoder False) or as its embedding (:573-577); the loop
    stops after the `eot_token` (:579).  Returns (tokens list[int], top-2 margin of the last-position logits per step)."""
    toks, margins = [], []
    with torch.no_grad():
        for _ in range(max_new_tokens):
            with autocast_ctx(autocast):
                logits = lm(idx if input_embeddings is None else None, input_embeddings, attention_mask)
            L = attention_mask.shape[-1]
            m = attention_mask.squeeze()
            a = torch.hstack([m, torch.zeros((L, 1)) + torch.finfo(logits.dtype).min])
            b = torch.vstack([a, torch.hstack([m[-1, :], torch.tensor([0])]).unsqueeze(0)])
            attention_mask = b.unsqueeze(0).unsqueeze(0)
            last = logits[:, -1]
            nxt = torch.argmax(last, dim=-1).reshape(-1, 1)
            top2 = last.float().topk(2, -1).values
            toks.append(int(nxt[0][0]))
            margins.append(float(top2[0, 0] - top2[0, 1]))
            if input_embeddings is not None:
                input_embeddings = torch.cat([input_embeddings, lm.model.embed_tokens(nxt)], dim=1)
            else:
                idx = torch.cat((idx, nxt), dim=1)
            if eot_token is not None and int(nxt[0][0]) == eot_token:
                break
    return toks, margins


# ------------------------------------------------------------------ MaskGIT parallel decoding
class TorchSampler:
    """The reference's randomness: torch.multinomial + uniform_()-based Gumbel noise on one generator
    (models/unigen.py:418-420, models/sampling.py:24-26,41-46).  Used to pin maskgit_generate_ref to the real
    reference's trajectory (tests/golden/g2_tiny_unigen.pt['maskgit'])."""

    def __init__(self, generator=None):
        self.generator = generator

    def sample(self, probs_flat):
        return torch.multinomial(probs_flat, 1, generator=self.generator)[:, 0]

    def gumbel(self, like):
        u = torch.zeros_like(like).uniform_(0, 1, generator=self.generator)
        return -torch.log((-torch.log(u.clamp(min=1e-20))).clamp(min=1e-20))


class InverseCdfSampler:
    """The product kernel's sampling rule (ug_maskgit_step): token = first index whose running sum of
    UNNORMALISED exp(logit - max) exceeds u * total, Gumbel noise = -log(-log(u2)), both from supplied uniforms
    u_sample[step] / u_conf[step] of shape [N, n]."""

    def __init__(self, u_sample, u_conf):
        self.u_sample, self.u_conf, self.step = u_sample, u_conf, 0

    def sample(self, probs_flat):
        u = self.u_sample[self.step].reshape(-1, 1).to(probs_flat.dtype)
        cdf = probs_flat.cumsum(-1)
        idx = (cdf <= u * cdf[:, -1:]).sum(-1)
        return idx.clamp(max=probs_flat.shape[-1] - 1)

    def gumbel(self, like):
        u = self.u_conf[self.step].to(like.dtype).reshape(like.shape)
        self.step += 1
        return -torch.log((-torch.log(u.clamp(min=1e-20))).clamp(min=1e-20))


def maskgit_generate_ref(lm, input_ids, uncond_input_ids, attention_mask, guidance_scale, temperature, timesteps,
                         schedule, n, text_vocab, mask_token_id, sampler, autocast=False, trace=None):
    """UniGen.t2i_generate (models/unigen.py:344-455): T rounds of {full forward on [cond; uncond], CFG on the
    code-book slice of the image positions, sample every position, keep known tokens, re-mask the
    max(1, min(#unknown - 1, floor(n * schedule((step+1)/T)))) least confident ones with Gumbel noise whose
    temperature is COMPOUNDED (temperature *= 1 - ratio)}; returns the last round's sampled ids."""
    embed = lm.model.embed_tokens
    cur_ids = input_ids[:, -(n + 1):-1].clone()
    emb = embed(input_ids)
    image_emb = emb[:, -(n + 1):-1]
    bsz = emb.shape[0]
    prefix, suffix = emb[:, :-(n + 1)], emb[:, -1:]
    cfg = guidance_scale > 1
    if cfg:
        prefix = torch.cat([prefix, embed(uncond_input_ids[:, :-(n + 1)])])
        suffix = torch.cat([suffix, suffix])
    sampled = None
    with torch.no_grad():
        for step in range(timesteps):
            img = torch.cat([image_emb, image_emb]) if cfg else image_emb
            seq = torch.cat([prefix, img, suffix], 1)
            with autocast_ctx(autocast):
                logits = lm(None, seq, attention_mask)
            lg = logits.float()[:, -(n + 1):-1, text_vocab:-1]
            if cfg:
                cond, uncond = lg[:bsz], lg[bsz:]
                lg = guidance_scale * (cond - uncond) + uncond
            probs = lg.softmax(-1)
            sampled = sampler.sample(probs.reshape(-1, probs.shape[-1])).view(bsz, n)
            unknown = cur_ids == mask_token_id
            sampled = torch.where(unknown, sampled, cur_ids)
            ratio = (step + 1) / timesteps
            sel = probs.gather(-1, sampled[..., None]).squeeze(-1)
            sel = torch.where(unknown, sel, torch.finfo(sel.dtype).max)
            mask_len = (n * schedule(torch.tensor(ratio))).floor().reshape(1, 1)
            mask_len = torch.max(torch.tensor([1]), torch.min(unknown.sum(-1, keepdim=True) - 1, mask_len))
            temperature = temperature * (1.0 - ratio)
            conf = torch.log(sel.clamp(min=1e-20)) + temperature * sampler.gumbel(sel)
            thr = torch.gather(torch.sort(conf, -1).values, 1, mask_len.long())
            masking = conf < thr
            if trace is not None:
                trace.append(dict(mixed=lg, sampled=sampled.clone(), masking=masking.clone(), conf=conf, thr=thr))
            image_emb = embed(torch.where(masking, mask_token_id, sampled + text_vocab))
            cur_ids = torch.where(masking, mask_token_id, sampled)
    return sampled
