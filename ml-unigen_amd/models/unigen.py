"""UniGen on the MI355X kernels -- drop-in for the reference's `models/unigen.py` (same class name,
constructor, method signatures and return conventions), so `training/train*.py` and
`evaluation/inference_*.py` run against it unchanged.  The backbone the reference builds from
transformers (`Qwen2ForCausalLM`, models/unigen.py:56-69) is `unigen_hip.modules.HipQwen2ForCausalLM`:
hand-written HIP kernels behind a C ABI, no torch math on the hot path.

Behavioural notes (all mirror the reference unless stated):
  * forward() with labels returns (logits, loss_t2i, loss_lm, loss_mmu); the three losses are the
    masked cross-entropies of models/unigen.py:310-338 (same slices, mean over non-ignored labels).
    `logits` is a lazy view that evaluates the lm_head only for the positions a caller slices
    (the reference materialises [B, L, 159867]); `label_smoothing` is accepted and ignored exactly
    like the reference (it never reaches F.cross_entropy).
  * precision: the reference's bf16-autocast training mode (fp32 master weights / residual stream,
    bf16 matmuls, fp32 statistics) is what the kernels implement, with or without an enclosing
    torch.autocast.
  * gen_proj_depth > 0 (alternate image embedding gen_embed -> gen_projector and the 8192-way img_head,
    unigen.py:74-92; off in every shipped config) is implemented for forward / training and MaskGIT
    generation; t2i_generate_ar with it raises.
"""
import json
import os
import time
import weakref
from typing import Optional

import torch

from unigen_hip import ops
from unigen_hip.lib import UniGenHipError
from unigen_hip.modules import (HipProjector, HipQwen2ForCausalLM, LazyLogits, _CrossEntropyFn, _HeadLossFn, _LinearFn,
                                _TableEmbedFn)
from unigen_hip.qwen2 import Qwen2Dims

from .modeling_utils import ConfigMixin, ModelMixin, register_to_config
from .sampling import cosine_schedule, mask_by_random_topk

# public Qwen2.5-1.5B(-Instruct) architecture (its config.json), used when no local HF directory exists
_KNOWN_LLM = {
    "qwen2.5-1.5b": dict(hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12,
                         num_key_value_heads=2, rope_theta=1000000.0, rms_norm_eps=1e-6, vocab_size=151936),
}


def _load_llm_config(llm_model_path, ckpt_base_path=""):
    """The reference reads this with AutoConfig.from_pretrained (models/unigen.py:52)."""
    cands = [llm_model_path]
    if ckpt_base_path:
        cands.insert(0, os.path.join(ckpt_base_path, os.path.basename(str(llm_model_path).rstrip("/"))))
    for c in cands:
        f = os.path.join(str(c), "config.json")
        if os.path.exists(f):
            with open(f) as fh:
                return json.load(fh)
    low = str(llm_model_path).lower()
    for key, cfg in _KNOWN_LLM.items():
        if key in low:
            return dict(cfg)
    raise UniGenHipError(f"cannot find an LLM config for '{llm_model_path}' (no config.json, unknown name)")


def _drop_anchor_from_state_dict(module, state_dict, prefix, local_metadata):
    module.llm.engine.fp.wait_pending_update()             # an overlapped optimizer update must land before anyone reads the tensors
    state_dict.pop(prefix + "_ddp_anchor", None)          # not part of the reference checkpoint format
    return state_dict


def _forgive_missing_anchor(module, incompatible_keys):
    incompatible_keys.missing_keys[:] = [k for k in incompatible_keys.missing_keys if not k.endswith("_ddp_anchor")]


class UniGen(ModelMixin, ConfigMixin):
    _supports_gradient_checkpointing = True

    @register_to_config
    def __init__(
            self,
            w_und_encoder: bool,
            vocab_size: int,
            llm_vocab_size: int,
            llm_model_path: str = '',
            codebook_size: int = 8192,
            num_vq_tokens: int = 256,
            load_from_pretrained: bool = True,
            mm_input_dim: int = 1024,
            gen_input_dim: int = 16,
            und_proj_depth: int = 0,
            gen_proj_depth: int = 0,
            use_gen_dim: bool = False,
            rope_theta: Optional[float] = None,
            scaling_factor: float = 1.0,
            rope_type: str = 'linear',
            vision_tower_name: Optional[str] = None,
            ckpt_base_path: str = "",
            **kwargs,
    ):
        super().__init__()
        device = kwargs.get("device", None) or torch.device("cuda", torch.cuda.current_device())
        self.vocab_size = vocab_size
        self.num_vq_tokens = num_vq_tokens
        llm_cfg = _load_llm_config(llm_model_path, ckpt_base_path)
        self.register_to_config(hidden_size=llm_cfg["hidden_size"])
        llm_cfg["vocab_size"] = vocab_size          # reference: config.vocab_size = vocab_size / resize_token_embeddings
        # reference :58-64: rope_theta / scaling_factor / rope_type override the config ONLY in the load_from_pretrained=True
        # branch (random init from config); with load_from_pretrained=False (HF weights) the reference ignores all three
        if load_from_pretrained and rope_theta is not None:
            llm_cfg["rope_theta"] = rope_theta
        if load_from_pretrained and scaling_factor != 1:
            llm_cfg["rope_scaling"] = {"factor": float(scaling_factor), "type": rope_type}
        dims = Qwen2Dims(**llm_cfg)
        seed = kwargs.get("init_seed", None)
        if seed is None:
            seed = int(torch.initial_seed() % (2 ** 31))
        elif seed < 0:              # caller initialises / loads the weights itself (checkpoints, device-side init)
            seed = None
        # load_from_pretrained=True in the reference means "random-init from config" (sic, unigen.py:58-65);
        # False means "load HF weights from llm_model_path".
        self.llm = HipQwen2ForCausalLM(dims, device, seed=seed)
        if not load_from_pretrained:
            self._load_hf_llm_weights(llm_model_path, ckpt_base_path)
        self.output_size = self.vocab_size
        self.img_output_size = codebook_size
        if gen_proj_depth > 0:
            # separate image-token embedding + MLP into the backbone and an 8192-way head out of it (reference :74-90)
            hidden = llm_cfg["hidden_size"]
            if use_gen_dim:
                self.gen_embed = torch.nn.Embedding(codebook_size + 1, gen_input_dim)
                layers, width = [torch.nn.Linear(gen_input_dim, hidden)], hidden
            else:
                self.gen_embed = torch.nn.Embedding(codebook_size + 1, hidden)
                layers, width = [torch.nn.Linear(hidden, hidden * 2)], hidden * 2
            for _ in range(1, gen_proj_depth):
                layers += [torch.nn.GELU(), torch.nn.Linear(width, hidden)]
                width = hidden
            self.gen_projector = HipProjector(*layers)
            self.img_head = torch.nn.Linear(hidden, codebook_size, bias=False)
            for m in (self.gen_embed, self.gen_projector, self.img_head):
                m.to(device)
            self.register_to_config(mask_token_id=codebook_size)
        else:
            self.register_to_config(mask_token_id=vocab_size - 1)
        self._loss_idx_cache = {}
        # Data parallelism (reference: accelerator.prepare wraps the model in DistributedDataParallel, train.py:492).
        # The backbone's parameters are views of one flat buffer whose gradients the kernels write directly and
        # unigen_hip.ddp.FlatGradSync averages, so DDP's reducer must leave them alone (see
        # `_ddp_params_and_buffers_to_ignore`); `_ddp_anchor` is the one ordinary parameter DDP always finds (it refuses
        # a module with none) and the input that puts the engine's autograd Functions on every graph.  It receives a
        # zero gradient per backward and never changes.
        eng = self.llm.engine
        self._ddp_anchor = torch.nn.Parameter(torch.zeros(1, device=eng.device))
        eng._anchor = self._ddp_anchor
        eng.extra_grad_params = self._ordinary_grad_params
        self.__dict__["_ddp_wrapper"] = None          # weakref to the DistributedDataParallel instance wrapping this model, if any
        self._register_state_dict_hook(_drop_anchor_from_state_dict)
        self.register_load_state_dict_post_hook(_forgive_missing_anchor)
        if w_und_encoder:
            if vision_tower_name is not None:
                self.init_vision_tower(vision_tower_name)
            self.add_mm_projector(max(2, und_proj_depth), mm_input_dim)

    # ------------------------------------------------------------------ data parallel plumbing
    def _flat_view_ids(self):
        return {id(p) for p in self.llm.engine.named_param_views().values()}

    @property
    def _ddp_params_and_buffers_to_ignore(self):
        """Read by torch's DistributedDataParallel constructor (nn/parallel/distributed.py: parameters_to_ignore): every
        name under which a flat-view parameter is reachable, including the tied `llm.lm_head.weight`.  Side-effect free:
        whether a wrapper exists is recorded by `_note_ddp_wrapper` below when DistributedDataParallel registers this model
        as its `.module`, and the flat weights are aligned to rank 0 at the first training forward (`TrainEngine._dp_sync`),
        which every rank reaches together."""
        flat = self._flat_view_ids()
        return [f"{mn}.{pn}" if mn else pn for mn, m in self.named_modules() for pn, p in m.named_parameters(recurse=False)
                if id(p) in flat]

    def _is_ddp_wrapped(self):
        """True while a live DistributedDataParallel instance holds this model: its reducer then averages the ordinary
        parameters (mm_projector, gen_*) and FlatGradSync only moves the flat buffer."""
        ref = self.__dict__.get("_ddp_wrapper")
        return ref is not None and ref() is not None

    def _ordinary_grad_params(self):
        if self._is_ddp_wrapped():
            return []
        flat = self._flat_view_ids()
        return [p for p in self.parameters() if id(p) not in flat and p is not self._ddp_anchor and p.requires_grad]

    def no_sync(self):
        """Context manager for gradient-accumulation micro-steps outside accelerate (inside it, `accelerator.accumulate`
        is honoured automatically): backward passes run here add to the local gradients and exchange nothing."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            eng = self.llm.engine
            prev, eng.require_grad_sync = eng.require_grad_sync, False
            try:
                yield
            finally:
                eng.require_grad_sync = prev
        return ctx()

    # ------------------------------------------------------------------ construction helpers
    def _load_hf_llm_weights(self, llm_model_path, ckpt_base_path):
        path = llm_model_path
        if ckpt_base_path and not os.path.exists(str(path)):
            path = os.path.join(ckpt_base_path, os.path.basename(str(llm_model_path).rstrip("/")))
        files = [f for f in os.listdir(path) if f.endswith(".safetensors")]
        if not files:
            raise UniGenHipError(f"no *.safetensors under {path}")
        from safetensors.torch import load_file
        sd = {}
        for f in files:
            sd.update(load_file(os.path.join(path, f)))
        own = self.llm.state_dict()
        V_ckpt = sd["model.embed_tokens.weight"].shape[0]
        with torch.no_grad():
            for k, v in sd.items():
                if k == "lm_head.weight" or k not in own:
                    continue
                if k == "model.embed_tokens.weight":      # == resize_token_embeddings(vocab_size) (unigen.py:68-69)
                    n = min(V_ckpt, own[k].shape[0])
                    own[k][:n].copy_(v[:n])
                else:
                    own[k].copy_(v)

    def _set_gradient_checkpointing(self, module, value=False):
        self.gradient_checkpointing = True

    def resize_token_embeddings(self, vocab_size):
        self.vocab_size = vocab_size
        self.llm.resize_token_embeddings(self.vocab_size)
        self.output_size = self.vocab_size

    def init_vision_tower(self, vision_tower_name):
        from .multimodal_encoder.builder import get_vision_tower
        self.register_to_config(vision_tower_name=vision_tower_name)
        if self.config.ckpt_base_path:
            vision_tower_name = os.path.join(self.config.ckpt_base_path, os.path.basename(vision_tower_name.rstrip("/")))
        self.vision_tower = get_vision_tower(vision_tower_name, freeze=False)

    def add_vision_tower(self, config):
        vision_tower_name = config.model.vision_tower.name
        self.init_vision_tower(vision_tower_name)
        vt = self.vision_tower
        mm_input_dim = vt.config.hidden_size if hasattr(vt, 'config') else vt.hidden_size
        self.add_mm_projector(config.model.unigen.get('und_proj_depth', 2), mm_input_dim)

    def add_mm_projector(self, mlp_depth, mm_input_dim):
        self.register_to_config(w_und_encoder=True)
        self.register_to_config(mm_input_dim=mm_input_dim)
        self.register_to_config(und_proj_depth=mlp_depth)
        hidden = self.config.hidden_size
        layers = [torch.nn.Linear(mm_input_dim, hidden)]
        for _ in range(1, mlp_depth):
            layers += [torch.nn.GELU(), torch.nn.Linear(hidden, hidden)]
        from unigen_hip.modules import HipProjector
        self.mm_projector = HipProjector(*layers).to(self.llm.engine.device)

    def _use_gen(self):
        return self.config.get('gen_proj_depth', 0) > 0

    def get_gen_embed(self, img_tokens):
        """gen_projector(gen_embed(img_tokens)) (reference :130-131); img_tokens are raw codes 0..codebook_size (the
        last id is this path's mask token)."""
        if not self._use_gen():
            raise UniGenHipError("get_gen_embed needs a model built with gen_proj_depth > 0")
        eng = self.llm.engine
        e = _TableEmbedFn.apply(self.gen_embed.weight, img_tokens.to(eng.device), eng.err_flag)
        return self.gen_projector(e)

    def prepare_inputs_for_t2i(self, input_ids, num_vq_tokens):
        emb = self.llm.model.embed_tokens(input_ids)
        if self._use_gen():                       # image slots carry the projected gen embeddings instead (reference :230-238)
            img = self.get_gen_embed(input_ids[:, -(num_vq_tokens + 1):-1].contiguous()).to(emb.dtype)
            emb = torch.cat([emb[:, :-(num_vq_tokens + 1)], img, emb[:, -1:]], dim=1)
        return emb

    def _img_head(self, hn):
        return _LinearFn.apply(hn, self.img_head.weight, None)

    def prepare_inputs_for_mmu(self, image_feats, spatial_shapes, input_ids, label_ids, prompt_template, input_ids_system=None):
        """Understanding-sample assembly for variable-size image features (reference models/unigen.py:133-228; callers
        training/train_w_clip_vit.py:754,801).  Row i is
            [system] <|im_start|><|mmu|><|soi|> | mm_projector(image_feats[i, :h_i*w_i]) | <|eoi|> input_ids[i, 1:] | pad...
        cut to `prompt_template.max_seq_len`; labels are ignore_id up to and including <|eoi|>, then label_ids[i, 1:],
        pads ignored; the key-validity mask is cut after the last `eos_token_id` label of the first row that has one (sic).  Returns (embeddings [B, L, H],
        attention_mask bool [B, max_seq_len], labels [B, L], input_ids_part1 [B, L1]).  One index computation for the
        whole batch instead of the reference's per-row concatenations; the projector and the embedding lookup run on the
        HIP kernels and stay differentiable."""
        pt = prompt_template
        dev = input_ids.device
        B, Lt = input_ids.shape
        N = image_feats.shape[1]
        pad_id, ignore, max_len = pt.text_tokenizer.pad_token_id, pt.ignore_id, pt.max_seq_len
        sp = pt.sptids_dict
        head = ['<|mmu|>', '<|im_start|>', '<|soi|>'] if pt.task_token_first else ['<|im_start|>', '<|mmu|>', '<|soi|>']
        part1 = torch.tensor([int(sp[t]) for t in head], dtype=torch.long, device=dev)[None].expand(B, 3)
        if input_ids_system is not None:
            part1 = torch.cat([input_ids_system.to(dev).long(), part1], dim=1)
        part1 = part1.contiguous()
        L1 = part1.shape[1]
        n_img = (spatial_shapes[:, 0] * spatial_shapes[:, 1]).to(dev).long()                      # [B]
        # row lengths before the cut: embeddings carry N trailing pads when training, (max_i n_img - n_img[i]) otherwise;
        # the label rows always carry N trailing ignore entries (reference :176-203)
        emb_len = L1 + n_img + Lt + (N if self.training else int(n_img.max()) - n_img)
        lab_len = L1 + n_img + Lt + N

        def common_length(lens, what):
            lo, hi = int(lens.min()), int(lens.max())
            if lo < max_len and lo != hi:
                raise UniGenHipError(f"prepare_inputs_for_mmu: {what} rows shorter than max_seq_len differ in length")
            return min(lo, max_len)
        T, T_lab = common_length(emb_len, "embedding"), common_length(lab_len, "label")
        const = lambda v: torch.full((), int(v), dtype=torch.long, device=dev)

        p = torch.arange(T, device=dev)[None]                                                     # [1, T]
        q = p - L1 - n_img[:, None]                                                               # offset into the text part
        is_img = (p >= L1) & (q < 0)
        text = torch.cat([const(sp['<|eoi|>']).expand(B, 1), input_ids[:, 1:].long()], dim=1)     # <|eoi|> replaces column 0
        ids = torch.where((q >= 0) & (q < Lt), text.gather(1, q.clamp(0, Lt - 1)), const(pad_id))
        head_ids = torch.nn.functional.pad(part1, (0, max(0, T - L1)), value=pad_id)[:, :T]
        ids = torch.where(p < L1, head_ids, ids)
        image_embeds = self.mm_projector(image_feats)                                             # [B, N, H]
        emb = self.llm.model.embed_tokens(ids)                                                    # image slots: pad rows, replaced below
        slot = (p - L1).clamp(0, N - 1).expand(B, T)
        img_rows = image_embeds.gather(1, slot[..., None].expand(B, T, image_embeds.shape[-1])).to(emb.dtype)
        full_embeddings = torch.where(is_img[..., None], img_rows, emb)

        if label_ids is None:
            label_ids = input_ids.clone()
        ql = torch.arange(T_lab, device=dev)[None] - L1 - n_img[:, None]
        labels = torch.where((ql >= 1) & (ql < Lt), label_ids.to(dev).long().gather(1, ql.clamp(0, Lt - 1)), const(ignore))
        labels = torch.where(labels == pad_id, const(ignore), labels)
        # key-validity mask (:211-226).  The reference walks the (row, column) list of eos labels with a cursor that never
        # moves past the first row it matched, so ONLY the first row that contains an eos label is cut (after its last
        # eos); every other row -- with or without eos labels -- stays fully valid.  Reproduced as is.
        is_eos = labels == pt.eos_token_id
        has = is_eos.any(1)
        first_row = torch.where(has.any(), has.long().argmax(), const(-1))
        last_eos = T_lab - 1 - is_eos.flip(-1).long().argmax(1)
        cut = torch.where(torch.arange(B, device=dev) == first_row, last_eos, const(T_lab - 1))
        attention_mask = torch.arange(max_len, device=dev)[None] <= cut[:, None]
        return full_embeddings, attention_mask, labels, part1

    # ------------------------------------------------------------------ forward
    def _loss_rows(self, B, L, bt, blm, bmmu, n, mode, device, lm_start=None):
        """Row indices (into [B*L]) of the logits each loss reads and of the labels it compares to
        (the slices of models/unigen.py:310-338)."""
        lm0 = bt if lm_start is None else lm_start
        key = (B, L, bt, blm, bmmu, n, mode, lm0)
        if key not in self._loss_idx_cache:
            def grid(b0, b1, p0, p1):
                b = torch.arange(b0, b1, device=device)[:, None] * L
                return (b + torch.arange(p0, p1, device=device)[None, :]).reshape(-1)
            segs = []
            if mode == 'mask':
                segs.append((grid(0, bt, L - (n + 1), L - 1), 0))
            else:
                segs.append((grid(0, bt, L - (n + 2), L - 1), 1))
            if blm > 0:
                segs.append((grid(lm0, lm0 + blm, 0, L - 1), 1))
            if bmmu > 0:
                segs.append((grid(B - bmmu, B, 0, L - 1), 1))
            bounds, r = [], 0
            for ix, _ in segs:
                bounds.append((r, r + ix.numel()))
                r += ix.numel()
            idx = torch.cat([ix for ix, _ in segs]) if r > 0 else torch.zeros(0, dtype=torch.long, device=device)
            lab = torch.cat([ix + s for ix, s in segs]) if r > 0 else idx
            self._loss_idx_cache[key] = (idx, lab, bounds)
        return self._loss_idx_cache[key]

    def forward(
            self,
            input_ids: torch.LongTensor,
            input_embeddings: Optional[torch.Tensor] = None,
            attention_mask: Optional[torch.Tensor] = None,
            labels: Optional[torch.LongTensor] = None,
            label_smoothing: float = 0.0,
            batch_size_t2i: int = 0,
            batch_size_lm: int = 0,
            batch_size_mmu: int = 0,
            max_seq_length: int = 128,
            num_vq_tokens: int = 256,
            t2i_mode: str = 'mask',
            **kwargs,
    ):
        eng = self.llm.engine
        gen = self._use_gen() and batch_size_t2i > 0
        if gen and input_embeddings is None:
            # the gen_projector path embeds the image slots through gen_embed + gen_projector; their ids are the raw
            # codes there (mask id = codebook_size), so the token-table lookup runs with those slots neutralised
            n = num_vq_tokens
            safe = input_ids.clone()
            safe[:, -(n + 1):-1] = 0
            emb = self.llm.model.embed_tokens(safe)
            img = self.get_gen_embed(input_ids[:, -(n + 1):-1].contiguous()).to(emb.dtype)
            input_embeddings = torch.cat([emb[:, :-(n + 1)], img, emb[:, -1:]], dim=1)
        if input_embeddings is None:
            out = self.llm.model(input_ids=input_ids, attention_mask=attention_mask)
        else:
            out = self.llm.model(inputs_embeds=input_embeddings, attention_mask=attention_mask)
        hn = out.last_hidden_state                                  # bf16 [B, L, H] (final norm applied)
        if gen:
            return self._forward_gen_head(hn, labels, batch_size_t2i, batch_size_lm, batch_size_mmu, num_vq_tokens, t2i_mode)
        if labels is None:
            return LazyLogits(eng, hn)          # stays on the autograd graph: DPO differentiates through its slices
        logits = LazyLogits(eng, hn.detach())
        B, L, _ = hn.shape
        idx, lab_idx, bounds = self._loss_rows(B, L, batch_size_t2i, batch_size_lm, batch_size_mmu, num_vq_tokens,
                                               t2i_mode, hn.device)
        nan = torch.full((), float("nan"), device=hn.device)
        if idx.numel() == 0:
            return logits, nan, 0., 0.
        lab = labels.to(hn.device).reshape(-1)[lab_idx].contiguous()
        live = [(s, b) for s, b in enumerate(bounds) if b[1] > b[0]]
        losses = _HeadLossFn.apply(eng._anchor, hn, eng, idx, lab, tuple(b for _, b in live))
        by_seg = {s: losses[j] for j, (s, _) in enumerate(live)}
        s = 0
        loss_t2i = by_seg.get(s, nan)
        s += 1
        loss_lm = 0.
        if batch_size_lm > 0:
            loss_lm = by_seg[s]
            s += 1
        loss_mmu = 0.
        if batch_size_mmu > 0:
            loss_mmu = by_seg[s]
        return logits, loss_t2i, loss_lm, loss_mmu

    def _forward_gen_head(self, hn, labels, bt, blm, bmmu, n, t2i_mode):
        """gen_proj_depth > 0 branch of forward (reference :255-341): the t2i rows go through img_head (codebook-wide
        logits, labels are raw codes), lm / mmu rows through the tied lm_head exactly as in the default branch; the first
        return value is img_logits."""
        eng = self.llm.engine
        B, L, _ = hn.shape
        img_logits = self._img_head(hn[:bt])                          # bf16 [bt, L, codebook]
        if labels is None:
            return img_logits
        labels = labels.to(hn.device)
        C = self.img_output_size
        if t2i_mode == 'mask':
            lg, lb = img_logits[:, -(n + 1):-1], labels[:bt, -(n + 1):-1]
        else:
            lg, lb = img_logits[:, -(n + 2):-1], labels[:bt, -(n + 1):]
        loss_t2i = _CrossEntropyFn.apply(lg.reshape(-1, C), lb.reshape(-1).contiguous())
        loss_lm, loss_mmu = 0., 0.
        if blm > 0 or bmmu > 0:
            idx, lab_idx, bounds = self._loss_rows(B, L, 0, blm, bmmu, n, t2i_mode, hn.device, lm_start=bt)
            lab = labels.reshape(-1)[lab_idx].contiguous()
            live = [(s, b) for s, b in enumerate(bounds) if b[1] > b[0]]
            losses = _HeadLossFn.apply(eng._anchor, hn, eng, idx, lab, tuple(b for _, b in live))
            by_seg = {s: losses[j] for j, (s, _) in enumerate(live)}
            s = 1
            if blm > 0:
                loss_lm = by_seg[s]
                s += 1
            if bmmu > 0:
                loss_mmu = by_seg[s]
        return img_logits, loss_t2i, loss_lm, loss_mmu

    # ------------------------------------------------------------------ MaskGIT generation
    @torch.no_grad()
    def t2i_generate(
            self,
            input_ids: Optional[torch.LongTensor] = None,
            uncond_input_ids: Optional[torch.LongTensor] = None,
            input_embeddings: Optional[torch.Tensor] = None,
            uncond_input_embeddings: Optional[torch.Tensor] = None,
            attention_mask: Optional[torch.Tensor] = None,
            temperature: float = 1.0,
            timesteps: int = 18,
            guidance_scale: int = 0,
            noise_schedule=cosine_schedule,
            generator: Optional[torch.Generator] = None,
            image_token_num_per_image: int = 256,
            text_vocab_size: int = 151936,
            **kwargs,
    ):
        """Iterative parallel decoding; step-for-step the procedure of reference models/unigen.py:344-455
        (incl. its quirks: untempered multinomial, compounded Gumbel temperature, >=1 token re-masked
        even on the last step, `sampled_ids` returned)."""
        n = image_token_num_per_image
        mask_token_id = self.config.mask_token_id
        embed = self.llm.model.embed_tokens
        cur_ids = input_ids[:, -(n + 1):-1].clone()
        gen = self._use_gen()                        # gen_projector path: raw codes in, gen embeddings, img_head out (:372-373)
        if input_embeddings is None:
            input_embeddings = embed(input_ids)
        image_embeddings = (self.get_gen_embed(cur_ids).to(input_embeddings.dtype) if gen
                            else input_embeddings[:, -(n + 1):-1])
        bsz = image_embeddings.shape[0]
        prefix = input_embeddings[:, :-(n + 1)]
        suffix = input_embeddings[:, -1:]
        cfg = guidance_scale > 1
        if cfg:
            un_prefix = (embed(uncond_input_ids[:, :-(n + 1)]) if uncond_input_embeddings is None
                         else uncond_input_embeddings[:, :-(n + 1)])
            prefix = torch.cat([prefix, un_prefix])
            suffix = torch.cat([suffix, suffix])
        sampled_ids = None
        eng = self.llm.engine
        L = prefix.shape[1] + n + 1
        seg_start = L - n - 2                        # <soi> | n image tokens | <eoi>
        # Prefix rows (padding + text) must not see the image segment for their keys / values to be round-invariant;
        # true for every mask the reference builds (create_attention_mask_predict_next), checked on the mask given.
        incremental = bool(kwargs.get("incremental", True)) and not torch.is_grad_enabled() and attention_mask is not None
        if incremental and isinstance(attention_mask, ops.MaskBits):
            # compressed mask (ops.mask_from_ids): no bit of a prefix row may be set at or beyond the segment's first column
            w0, sh = seg_start // 64, seg_start % 64
            words = attention_mask.bits[:, :seg_start, w0:]
            keep = torch.full((words.shape[-1],), -1, dtype=torch.int64, device=words.device)
            keep[0] = -1 << sh                           # the straddling word: only columns >= seg_start count
            incremental = not bool(((words & keep) != 0).any())
        elif incremental:
            incremental = torch.is_tensor(attention_mask) and attention_mask.dim() == 4 \
                and not bool((attention_mask[:, 0, :seg_start, seg_start:] == 0).any())
        sess = None
        trace = kwargs.get("trace", None)
        for step in range(timesteps):
            img = torch.cat([image_embeddings, image_embeddings]) if cfg else image_embeddings
            if incremental:
                R = img.shape[0]
                if sess is None:
                    seq = torch.cat([prefix, img, suffix], 1).float()
                    mb = eng.mask_bits(attention_mask, R, L)
                    sess, hn = eng.maskgit_begin(seq.reshape(R * L, -1).contiguous(), mb, L, seg_start)
                    eng.check_errors()
                else:
                    seg = torch.cat([prefix[:, -1:], img, suffix], 1).float()
                    hn = eng.maskgit_step(sess, seg.reshape(R * (n + 2), -1).contiguous())
                rows = hn.view(R, n + 2, -1)[:, 1:n + 1].reshape(R * n, -1).contiguous()
                lg = (self._img_head(rows) if gen else eng.head_slice(rows, text_vocab_size, self.vocab_size - 1)).reshape(R, n, -1)
            elif gen:
                seq = torch.cat([prefix, img, suffix], 1)
                hn = self.llm.model(inputs_embeds=seq, attention_mask=attention_mask).last_hidden_state
                lg = self._img_head(hn[:, -(n + 1):-1].contiguous())
            else:
                seq = torch.cat([prefix, img, suffix], 1)
                out = self(input_ids=input_ids, input_embeddings=seq, attention_mask=attention_mask)
                # only the image positions x codebook columns are ever read (reference slices the dense logits)
                lg = out[:, -(n + 1):-1, text_vocab_size:-1]
            ratio = 1.0 * (step + 1) / timesteps
            mask_len = int(torch.floor(n * noise_schedule(torch.tensor(ratio))).item())
            temperature = temperature * (1.0 - ratio)
            # one fused step on the device: CFG mix, softmax, categorical draw (inverse CDF on uniforms from `generator`),
            # confidence + Gumbel noise, re-masking of the max(1, min(#unknown - 1, mask_len)) least confident positions
            u_dev = lg.device if generator is None else generator.device
            u = torch.rand((2, bsz, n), device=u_dev, generator=generator).to(lg.device)
            sampled_ids, cur_ids, next_ids = ops.maskgit_step(lg.contiguous(), bsz, n, cfg, guidance_scale, u[0], u[1], cur_ids,
                                                              mask_token_id, 0 if gen else text_vocab_size, mask_len, temperature)
            if trace is not None:                        # parity tests follow the trajectory round by round
                trace.append((sampled_ids.clone(), next_ids.clone()))
            image_embeddings = self.get_gen_embed(next_ids).to(input_embeddings.dtype) if gen else embed(next_ids)
        return sampled_ids

    # ------------------------------------------------------------------ autoregressive generation
    def drop_decode_session(self):
        """Release the decode step kept from the last `t2i_generate_ar` call: the static KV cache of every layer (rows x (prefix + n)
        tokens), the decode scratch, and the captured graph with its private memory pool -- hundreds of MB that otherwise stay
        allocated until a call with different shapes replaces them.  `train(True)` calls this (periodic evaluation inside a training
        run must not keep generation buffers for the rest of it); `UNIGEN_AR_GRAPH_CACHE=0` disables keeping a session at all."""
        eng = getattr(getattr(self, "llm", None), "engine", None)
        if eng is not None:
            eng._ar_session = None

    def train(self, mode: bool = True):
        if mode:
            self.drop_decode_session()
        return super().train(mode)

    @torch.no_grad()
    def t2i_generate_ar(
            self,
            input_ids: Optional[torch.LongTensor] = None,
            uncond_input_ids: Optional[torch.LongTensor] = None,
            input_embeddings: Optional[torch.Tensor] = None,
            uncond_input_embeddings: Optional[torch.Tensor] = None,
            attention_mask: Optional[torch.Tensor] = None,
            guidance_scale: int = 0,
            temperature: float = 1.0,
            text_vocab_size: int = 151936,
            image_token_num_per_image: int = 256,
            generator: Optional[torch.Generator] = None,
            **kwargs,
    ):
        """Token-by-token image generation with CFG (reference models/unigen.py:457-521).  The reference
        only works when both embedding tensors are supplied; ids are accepted here too
        and embedded, which is what its callers intend."""
        from unigen_hip.qwen2 import DecodeState
        gen = self._use_gen()          # gen_projector path (reference :486-495,512-514): img_head on the last hidden state, the next
        n = image_token_num_per_image  # input is gen_projector(gen_embed(raw code)); no text-vocabulary offset anywhere
        embed = self.llm.model.embed_tokens
        eng = self.llm.engine
        dev = eng.device
        if input_embeddings is None:
            input_embeddings = embed(input_ids)
        if uncond_input_embeddings is None:
            uncond_input_embeddings = embed(uncond_input_ids)
        bsz = input_embeddings.shape[0]
        prefix = torch.cat([input_embeddings[:, :-(n + 1)], uncond_input_embeddings[:, :-(n + 1)]]).float()
        R, P, _ = prefix.shape
        key_valid = None
        if attention_mask is not None:
            if attention_mask.dim() != 2:
                raise UniGenHipError("t2i_generate_ar expects the 2-D [rows, L] attention mask the reference slices")
            key_valid = attention_mask[:, :P].to(dev) != 0
        greedy = bool(kwargs.get("greedy", False))          # argmax instead of multinomial: deterministic parity tests
        logit_trace = kwargs.get("trace")                   # optional list: fp32 [rows, V] head logits of every eager step
        use_graph = bool(kwargs.get("use_graph", True))
        code_lo, code_hi = text_vocab_size, self.vocab_size - 1          # logits[..., text_vocab_size:-1]
        V = code_hi - code_lo
        fused = (R <= 32 and eng.dims.hidden_size >= 256 and eng.dims.hidden_size % 32 == 0 and not kwargs.get("torch_sampler", False)
                 and not gen)
        # The captured decode step is kept ACROSS calls (round 5): Best-of-N generation calls this method once per prompt with the
        # same shapes (evaluation/inference_unigen_cot.py:318-331), and capturing costs ~6 ms of a 330 ms call (an eager warm-up
        # step + the capture).  A session = every buffer the graph reads or writes (cache, position, accumulators, token /
        # embedding slots, uniforms) + the graph; it is reused only when every size, every sampling constant baked into a kernel
        # argument and the weight storage are the same (UNIGEN_AR_GRAPH_CACHE=0 turns the reuse off), and dropped on any error.
        sess_key = (R, P, n, bsz, V, int(text_vocab_size), greedy, float(guidance_scale), float(temperature), key_valid is None, str(dev),
                    bool(getattr(eng, "decode_fused", True)), os.environ.get("UNIGEN_DECODE_SW", "1"),
                    eng.fp.w("embed").data_ptr(), eng.fp.w("l0.wqkv").data_ptr(), eng.fp.p("embed").data_ptr(), eng.fp.p("norm").data_ptr(),
                    # (every other pointer the captured step bakes in lives in the same two flat buffers; the last layer's weights and the
                    # RoPE tables stand in for "nothing was reallocated in between")
                    eng.fp.w(f"l{eng.dims.num_hidden_layers - 1}.wdown").data_ptr(), tuple(t.data_ptr() for t in eng.rope(P + n)))
        sess = getattr(eng, "_ar_session", None) if (use_graph and fused and os.environ.get("UNIGEN_AR_GRAPH_CACHE", "1") != "0") else None
        if sess is not None and sess["key"] != sess_key:
            sess = None
        eng._ar_session = None                       # (put back at the end of a call that completed)
        if sess is not None:
            st, out_tokens, x, tok = sess["st"], sess["out_tokens"], sess["x"], sess["tok"]
            if key_valid is not None:
                st.key_valid[:, :P].copy_(key_valid)
        else:
            st = DecodeState(eng.dims, R, P + n, dev, key_valid=key_valid)
            out_tokens = torch.zeros((bsz, n), dtype=torch.int, device=dev)
            x = torch.empty((R, eng.dims.hidden_size), dtype=torch.float32, device=dev)      # static: next token's embedding
            tok = torch.zeros((bsz, 1), dtype=torch.long, device=dev)                        # static: last sampled token
        if fused:
            # lm-head as a weight-streaming GEMV into a raw fp32 accumulator + ONE sampling kernel per step (CFG mix,
            # temperature, softmax, inverse-CDF draw on uniforms taken from `generator` up front, next input embedding)
            u_dev = dev if generator is None else generator.device
            fresh = None if greedy else torch.rand((n, bsz), device=u_dev, generator=generator).to(dev)
            if sess is not None:
                acc_head, uniforms = sess["acc_head"], sess["uniforms"]
                if uniforms is not None:
                    uniforms.copy_(fresh)
            else:
                acc_head = torch.zeros((R, V), dtype=torch.float32, device=dev)
                uniforms = fresh
            w_head = eng.fp.w("embed")[code_lo:code_hi]
            w_embed = eng.fp.p("embed")

            def keep_logits():                     # parity tests follow the head's raw logits step by step (eager runs only)
                if logit_trace is not None and not torch.cuda.is_current_stream_capturing():
                    logit_trace.append(acc_head.clone())

            def sample(hn):
                ops.decode_gemv_(hn, w_head, acc_head)
                keep_logits()
                ops.ar_sample_(acc_head, bsz, V, guidance_scale, temperature, greedy, uniforms, st.pos, P, n, w_embed,
                               text_vocab_size, tok, out_tokens, x)
        else:
            def sample(hn):
                # (gen path: the reference mixes the bf16 img_head outputs in bf16 under autocast, :498-500)
                lg = self._img_head(hn) if gen else eng.head_slice(hn, code_lo, code_hi).float()
                cond, uncond = lg[:bsz], lg[bsz:]
                lg = (uncond + guidance_scale * (cond - uncond)).float()
                if greedy:
                    nxt = lg.argmax(-1, keepdim=True)
                else:
                    nxt = torch.multinomial(torch.softmax(lg / temperature, dim=-1), num_samples=1, generator=generator)
                tok.copy_(nxt)
                if gen:
                    x.copy_(self.get_gen_embed(torch.cat([nxt, nxt]))[:, 0])
                else:
                    x.copy_(embed(torch.cat([nxt, nxt]) + text_vocab_size)[:, 0])

        timing = kwargs.get("timing")               # optional dict: wall seconds per phase (adds host syncs; measurement runs only)

        def mark(name, t0=[None]):
            if timing is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                if t0[0] is not None:
                    timing[name] = timing.get(name, 0.0) + now - t0[0]
                t0[0] = now

        mark("setup")
        sample(eng.prefill(st, prefix, key_valid))         # (capturing the prefill too was measured: 10.3 vs 10.6 ms, it is GPU-bound at 2 208 tokens)
        mark("prefill")
        if not fused:
            out_tokens[:, 0] = tok[:, 0]

        # single-writer layer (csrc/decode_sw.hip): the final RMSNorm and the head slice are ONE launch behind the last layer
        sw_head = fused and eng.decode_sw(st)

        def step():
            if sw_head:
                eng.decode_step_logits(st, x, w_head, acc_head)            # (also advances the cache position)
                keep_logits()
                ops.ar_sample_(acc_head, bsz, V, guidance_scale, temperature, greedy, uniforms, st.pos, P, n, w_embed,
                               text_vocab_size, tok, out_tokens, x)
                return
            hn = eng.decode_step(st, x)            # (also advances the cache position)
            sample(hn)

        graph = sess["graph"] if sess is not None else None
        for i in range(1, n):
            if graph is None and use_graph and (generator is None or fused) and i == 2:
                # step 1 ran eagerly (warm-up: allocations, lazy inits); capture step 2 and replay it from then on
                mark("eager_step")
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    step()
                mark("capture")
                graph.replay()                      # capture only records: this replay IS step 2
            elif graph is not None:
                graph.replay()
            else:
                step()
            if not fused:
                out_tokens[:, i] = tok[:, 0]
        mark("replay")
        eng.last_decode_graph = graph is not None
        if graph is not None and fused and os.environ.get("UNIGEN_AR_GRAPH_CACHE", "1") != "0":
            eng._ar_session = {"key": sess_key, "st": st, "out_tokens": out_tokens, "x": x, "tok": tok, "acc_head": acc_head,
                               "uniforms": uniforms, "graph": graph}
            return out_tokens.clone()                # (the session's buffer is overwritten by the next call)
        return out_tokens

    # ------------------------------------------------------------------ plain causal generation
    @torch.no_grad()
    def generate(self, input_ids=None, input_embeddings=None, attention_mask=None, max_new_tokens=20, do_sample=False,
                 temperature=1.0, top_k=None, top_p=None, eos_token_id=None, pad_token_id=None, use_cache=True,
                 generator=None, **kwargs):
        """Causal text generation with the conventions of transformers' `generate`, which the reference delegates to
        (models/unigen.py:584-588; caller evaluation/inference_unigen_cot.py:360): prompts as ids [B, L] or as
        `input_embeddings` [B, L, H] with an optional 2-D [B, L] key-validity mask (left padding); greedy when
        `do_sample` is false, otherwise temperature -> top-k -> top-p -> multinomial; a row that produced
        `eos_token_id` is filled with `pad_token_id` from then on and decoding stops when every row has finished.
        Returns prompt + continuation [B, L + new] for ids, the continuation alone [B, new] for embeddings (HF rule).
        One prefill into the static KV cache, then one decode step per token (`use_cache` is accepted and ignored: the
        recompute form would return the same tokens)."""
        from unigen_hip.qwen2 import DecodeState
        from .sampling import top_k_top_p_filtering
        unsupported = [k for k in ("num_beams", "num_return_sequences", "repetition_penalty", "penalty_alpha") if kwargs.get(k) not in (None, 1, 1.0)]
        if unsupported:
            raise UniGenHipError(f"generate: {unsupported} are not implemented (greedy / sampling only)")
        if "max_length" in kwargs and kwargs["max_length"] is not None and input_ids is not None:
            max_new_tokens = int(kwargs["max_length"]) - input_ids.shape[1]
        eng = self.llm.engine
        embed = self.llm.model.embed_tokens
        dev = eng.device
        prompt = (embed(input_ids.to(dev)) if input_embeddings is None else input_embeddings.to(dev)).float()
        R, L, _ = prompt.shape
        key_valid = None
        if attention_mask is not None:
            if attention_mask.dim() != 2:
                raise UniGenHipError("generate expects a 2-D [rows, L] attention mask (1 = real token)")
            key_valid = attention_mask.to(dev) != 0
        eos = [] if eos_token_id is None else ([int(e) for e in eos_token_id] if isinstance(eos_token_id, (list, tuple)) else [int(eos_token_id)])
        if eos and pad_token_id is None:
            pad_token_id = eos[0]
        st = DecodeState(eng.dims, R, L + max_new_tokens, dev, key_valid=key_valid)
        hn = eng.prefill(st, prompt, key_valid)
        eng.check_errors()
        V = self.config.vocab_size
        x = torch.empty((R, eng.dims.hidden_size), dtype=torch.float32, device=dev)
        out = torch.full((R, max_new_tokens), int(pad_token_id or 0), dtype=torch.long, device=dev)
        done = torch.zeros(R, dtype=torch.bool, device=dev)
        n_out = 0
        for i in range(max_new_tokens):
            last = eng.head_slice(hn, 0, V).float()
            if do_sample:
                if temperature is not None and temperature != 1.0:
                    last = last / temperature
                last = top_k_top_p_filtering(last, top_k=int(top_k or 0), top_p=float(1.0 if top_p is None else top_p))
                u_dev = dev if generator is None else generator.device
                nxt = torch.multinomial(torch.softmax(last, dim=-1).to(u_dev), num_samples=1, generator=generator).to(dev)
            else:
                nxt = last.argmax(-1, keepdim=True)
            if eos:
                nxt = torch.where(done[:, None], torch.full_like(nxt, int(pad_token_id)), nxt)
            out[:, i] = nxt[:, 0]
            n_out = i + 1
            if eos:
                done |= torch.isin(nxt[:, 0], torch.tensor(eos, device=dev))
                if bool(done.all()):
                    break
            if i + 1 < max_new_tokens:
                x.copy_(embed(nxt)[:, 0])
                hn = eng.decode_step(st, x)            # (also advances the cache position)
        out = out[:, :n_out]
        if input_embeddings is None:
            return torch.cat([input_ids.to(dev), out], dim=1)
        return out

    # ------------------------------------------------------------------ text decoding for understanding
    @torch.no_grad()
    def mmu_generate(self, idx=None, input_embeddings=None, attention_mask=None, max_new_tokens=100, temperature=1.0,
                     top_k=None, eot_token=None, use_cache=True):
        """Greedy / top-k text continuation (reference models/unigen.py:523-581).  The reference re-runs the whole
        growing sequence every step and extends the additive mask by one row that copies the previous last row;
        here the prompt is prefilled once under its mask into the static KV cache and every new token is one decode
        step that attends to the keys the prompt's last row could see plus everything generated since (the same
        function of the inputs; `use_cache=False` keeps the step-by-step recomputation for comparison)."""
        if use_cache and attention_mask is not None and attention_mask.shape[0] == 1:
            return self._mmu_generate_cached(idx, input_embeddings, attention_mask, max_new_tokens, temperature, top_k, eot_token)
        return self._mmu_generate_recompute(idx, input_embeddings, attention_mask, max_new_tokens, temperature, top_k, eot_token)

    @staticmethod
    def _pick_next(last, temperature, top_k):
        if temperature > 0:
            last = last / temperature
            if top_k is not None:
                v, _ = torch.topk(last, min(top_k, last.size(-1)))
                last[last < v[:, [-1]]] = -float('Inf')
            return torch.multinomial(torch.softmax(last, dim=-1), num_samples=1)
        return torch.argmax(last, dim=-1).reshape(-1, 1)

    @torch.no_grad()
    def _mmu_generate_cached(self, idx, input_embeddings, attention_mask, max_new_tokens, temperature, top_k, eot_token):
        from unigen_hip.qwen2 import DecodeState
        eng = self.llm.engine
        embed = self.llm.model.embed_tokens
        prompt = (embed(idx) if input_embeddings is None else input_embeddings).float()
        dev = prompt.device
        L = prompt.shape[1]
        mb = eng.mask_bits(attention_mask, 1, L)
        eng.check_errors()
        key_valid = (attention_mask.reshape(L, L)[-1] == 0).view(1, L)
        st = DecodeState(eng.dims, 1, L + max_new_tokens, dev, key_valid=key_valid)
        hn = eng.prefill(st, prompt, mask_bits=mb)
        V = self.config.vocab_size
        x = torch.empty((1, eng.dims.hidden_size), dtype=torch.float32, device=dev)
        result = []
        for i in range(max_new_tokens):
            last = eng.head_slice(hn, 0, V).float()
            idx_next = self._pick_next(last, temperature, top_k)
            result.append(idx_next[0][0])
            if eot_token is not None and idx_next.cpu() == eot_token:
                break
            if i + 1 < max_new_tokens:
                x.copy_(embed(idx_next)[:, 0])
                hn = eng.decode_step(st, x)            # (also advances the cache position)
        return result

    @torch.no_grad()
    def mmu_generate_batch(self, idx=None, input_embeddings=None, attention_mask=None, max_new_tokens=100, temperature=0.0,
                           top_k=None, eot_token=None):
        """`mmu_generate` for up to 32 prompts at once -- the rating loop of CoT-V (reference
        evaluation/inference_unigen_cot.py:308-415 calls mmu_generate once per (image, question) pair; every decode
        step streams the whole backbone whatever the row count, so R pairs cost about one).  Rows are LEFT-padded to a
        common length L: idx [R, L] (or input_embeddings [R, L, H]) and the rows' dense additive masks [R, 1, L, L] with
        the pad columns blocked (the reference's mask builders do that for left-padded rows).  Each row follows the
        procedure of `mmu_generate`: prefill under its mask, then one decode step per token attending to the keys its
        last prompt row could see plus everything generated since.  Returns R lists of tokens, each cut after its
        `eot_token`."""
        from unigen_hip.qwen2 import DecodeState
        eng = self.llm.engine
        embed = self.llm.model.embed_tokens
        prompt = (embed(idx) if input_embeddings is None else input_embeddings).float()
        dev = prompt.device
        R, L = prompt.shape[0], prompt.shape[1]
        if R > 32:
            raise ValueError("mmu_generate_batch: at most 32 rows per call")
        if attention_mask is None or tuple(attention_mask.shape) != (R, 1, L, L):
            raise ValueError("mmu_generate_batch: attention_mask must be the rows' dense [R, 1, L, L] additive masks")
        mb = eng.mask_bits(attention_mask, R, L)
        eng.check_errors()
        key_valid = attention_mask[:, 0, -1, :] == 0
        st = DecodeState(eng.dims, R, L + max_new_tokens, dev, key_valid=key_valid)
        hn = eng.prefill(st, prompt, mask_bits=mb)
        V = self.config.vocab_size
        x = torch.empty((R, eng.dims.hidden_size), dtype=torch.float32, device=dev)
        tokens = torch.zeros((R, max_new_tokens), dtype=torch.long, device=dev)
        done = torch.zeros(R, dtype=torch.bool, device=dev)
        lengths = torch.full((R,), max_new_tokens, dtype=torch.long, device=dev)
        for i in range(max_new_tokens):
            last = eng.head_slice(hn, 0, V).float()
            idx_next = self._pick_next(last, temperature, top_k)
            tokens[:, i] = idx_next[:, 0]
            if eot_token is not None:
                hit = (idx_next[:, 0] == eot_token) & ~done
                lengths = torch.where(hit, torch.full_like(lengths, i + 1), lengths)
                done |= hit
                if bool(done.all()):
                    break
            if i + 1 < max_new_tokens:
                x.copy_(embed(idx_next)[:, 0])
                hn = eng.decode_step(st, x)            # (also advances the cache position)
        tokens, lengths = tokens.cpu(), lengths.cpu()
        return [list(tokens[r, :int(lengths[r])]) for r in range(R)]

    def _mmu_generate_recompute(self, idx, input_embeddings, attention_mask, max_new_tokens, temperature, top_k, eot_token):
        device = idx.device if idx is not None else input_embeddings.device
        result = []
        neg = torch.finfo(torch.bfloat16).min
        for _ in range(max_new_tokens):
            logits = self(idx, input_embeddings=input_embeddings, attention_mask=attention_mask)
            last = logits[:, -1, :].float()
            L = attention_mask.shape[-1]
            m = attention_mask.reshape(L, L)
            grown = torch.full((L + 1, L + 1), float(neg), device=m.device, dtype=m.dtype)
            grown[:L, :L] = m
            grown[L, :L] = m[-1]
            grown[L, L] = 0
            attention_mask = grown[None, None]
            if temperature > 0:
                last = last / temperature
                if top_k is not None:
                    v, _ = torch.topk(last, min(top_k, last.size(-1)))
                    last[last < v[:, [-1]]] = -float('Inf')
                idx_next = torch.multinomial(torch.softmax(last, dim=-1), num_samples=1)
            else:
                idx_next = torch.argmax(last, dim=-1).reshape(-1, 1)
            result.append(idx_next[0][0])
            if self.config.w_und_encoder:
                input_embeddings = torch.cat([input_embeddings, self.llm.model.embed_tokens(idx_next)], dim=1)
            else:
                idx = torch.cat((idx, idx_next), dim=1)
            if eot_token is not None and idx_next.cpu() == eot_token:
                break
        return result


def _note_ddp_wrapper(parent, name, sub):
    """Global module-registration hook: `DistributedDataParallel.__init__` does `self.module = module` (reference:
    accelerator.prepare, training/train.py:492).  The weak reference dies with the wrapper, so a model that is unwrapped
    again goes back to averaging its ordinary parameters itself."""
    if isinstance(sub, UniGen) and isinstance(parent, torch.nn.parallel.DistributedDataParallel):
        sub.__dict__["_ddp_wrapper"] = weakref.ref(parent)
    return None


torch.nn.modules.module.register_module_module_registration_hook(_note_ddp_wrapper)
