"""Fused AdamW over flat parameter runs (replaces torch.optim.AdamW.step at reference
training/train.py:324-330,780; same constructor / param-group interface, same update rule).

Parameters that are adjacent views of one buffer (the UniGen backbone's flat fp32 master/grad
buffers) are merged into runs, so a 1.56 B-parameter step is a few dozen launches of one
HBM-bound kernel instead of one launch per tensor."""
import torch

from . import ops


class _Run:
    __slots__ = ("params", "p_ptr", "numel", "m", "v", "group")


class FusedAdamW(torch.optim.Optimizer):
    """State layout = torch.optim.AdamW's: `self.state[p]` holds `step`, `exp_avg`, `exp_avg_sq` per parameter, so
    `state_dict()` / `load_state_dict()` (what the reference checkpoints through `accelerator.save_state` and
    utils/checkpoint.py:67-69) interoperate with torch.optim.AdamW.  The moment tensors are views of one buffer per
    flat run, which is what the kernel walks."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, overlap=False):
        """overlap=True: the update of the backbone's flat buffers (HBM-bound, ~9 ms for 1.56 B parameters) is issued on a
        side stream, so whatever the caller runs next that does not touch the backbone -- the frozen MAGVITv2 tokenisation
        of the next batch is MFMA-bound and comes first in every training script -- runs beside it.  The engine makes its own
        streams wait for the update before the first read or write of weights / gradients (embedding lookup, refresh of the
        bf16 copies, gradient clearing, state_dict); code that reads backbone parameters or gradients through plain torch
        ops between step() and the next forward must call `synchronize()` first.  Ordinary parameters (mm_projector, ...)
        are always updated on the current stream."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._runs = None
        self._step = 0
        self.overlap = bool(overlap)
        self.overlap_blocks = 256                  # one small workgroup per CU: the overlapped update yields registers / LDS
        self._side = None
        self._pending = None

    def synchronize(self):
        """Make the current stream wait for an overlapped update still in flight."""
        if self._pending is not None:
            torch.cuda.current_stream().wait_event(self._pending)
            self._pending = None

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            self.synchronize()                     # an in-place clear would race with the update reading the gradients
        return super().zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        self.synchronize()
        return super().state_dict()

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._runs = None                          # rebuilt (moments carried over from self.state) at the next step

    def _build_runs(self):
        runs = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.requires_grad]
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise ops._l.UniGenHipError("FusedAdamW needs contiguous fp32 parameters in GPU memory")
            ps.sort(key=lambda t: t.data_ptr())
            cur = None
            for p in ps:
                end = None if cur is None else cur.p_ptr + cur.numel * 4
                same_store = cur is not None and cur.params[-1].untyped_storage().data_ptr() == p.untyped_storage().data_ptr()
                if cur is not None and same_store and 0 <= p.data_ptr() - end <= 1024 and (p.data_ptr() - cur.p_ptr) % 16 == 0:
                    cur.params.append(p)
                    cur.numel = (p.data_ptr() - cur.p_ptr) // 4 + p.numel()
                else:
                    cur = _Run()
                    cur.params, cur.p_ptr, cur.numel, cur.group = [p], p.data_ptr(), p.numel(), gi
                    runs.append(cur)
        for r in runs:
            dev = r.params[0].device
            r.m = torch.zeros(r.numel, dtype=torch.float32, device=dev)
            r.v = torch.zeros(r.numel, dtype=torch.float32, device=dev)
            for p in r.params:
                o = (p.data_ptr() - r.p_ptr) // 4
                m, v = r.m[o:o + p.numel()].view_as(p), r.v[o:o + p.numel()].view_as(p)
                old = self.state.get(p)
                if old:                            # loaded checkpoint / earlier run layout: carry the moments over
                    m.copy_(old["exp_avg"])
                    v.copy_(old["exp_avg_sq"])
                    self._step = max(self._step, int(float(old.get("step", 0))))
                self.state[p] = {"step": torch.tensor(float(self._step)), "exp_avg": m, "exp_avg_sq": v}
        self._runs = runs

    def load_state_dict(self, state_dict):
        """torch.optim.AdamW layout in (from this class or from torch.optim.AdamW); the loaded moments are copied into the
        flat run buffers and the single bias-correction step count resumes from the stored one."""
        super().load_state_dict(state_dict)
        self._step = 0
        self._runs = None
        with torch.no_grad():
            self._build_runs()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        if self._runs is None:
            self._build_runs()
        self._step += 1
        lib = ops._l.load()
        # masters whose bf16 compute mirror is in sync right now stay in sync: the kernel writes the mirror too
        synced = {}
        main = torch.cuda.current_stream()
        side_used = False
        for r in self._runs:
            g = self.param_groups[r.group]
            first = r.params[0]
            if first.grad is None:
                if any(p.grad is not None for p in r.params):
                    raise ops._l.UniGenHipError("FusedAdamW: partially missing gradients inside a flat run")
                continue
            g_ptr = first.grad.data_ptr()
            for p in r.params[1:]:
                if p.grad is None or p.grad.data_ptr() - g_ptr != p.data_ptr() - r.p_ptr:
                    raise ops._l.UniGenHipError("FusedAdamW: gradient views do not mirror the parameter layout")
            b1, b2 = g["betas"]
            owner, mirror = ops.find_bf16_mirror(r.p_ptr, r.numel)
            if owner is not None:
                if id(owner) not in synced:
                    synced[id(owner)] = (owner, owner._seen_version == owner.master._version)
                if not synced[id(owner)][1]:
                    mirror = 0
            stream = main
            if self.overlap and owner is not None:          # flat backbone buffers only: every reader goes through the engine
                if self._side is None:
                    self._side = torch.cuda.Stream()
                if not side_used:
                    self._side.wait_stream(main)
                    side_used = True
                stream = self._side
            rc = lib.ug_adamw_flat(r.p_ptr, g_ptr, r.m.data_ptr(), r.v.data_ptr(), mirror, r.numel, float(g["lr"]), b1, b2,
                                   g["eps"], g["weight_decay"], self._step, float(grad_scale),
                                   self.overlap_blocks if stream is not main else 0, stream.cuda_stream)
            ops._l.check(rc, "ug_adamw_flat")
            # the kernel wrote through a raw pointer: bump the (shared) version counter so the engine
            # knows its bf16 compute copies are stale
            torch.autograd.graph.increment_version(first)
        if side_used:
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._pending = ev
            for owner, _ in synced.values():
                owner.pending_update = ev
        for owner, ok in synced.values():
            if ok:
                owner._seen_version = owner.master._version
        for st in self.state.values():
            st["step"].fill_(float(self._step))
        return loss
