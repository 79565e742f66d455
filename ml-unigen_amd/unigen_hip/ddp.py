"""Data-parallel gradient exchange for the flat gradient buffer (replaces the DistributedDataParallel reducer that
`accelerator.prepare` / `accelerator.backward` put behind the reference's step, training/train.py:492,775;
SURVEY.md §2.2, §8e).

One process per GPU, full replica.  The only collective on the training path is the MEAN all-reduce of the flat
gradient buffer, issued bucket by bucket on a side HIP stream while backward is still retiring earlier layers:

    backward (main stream)   ... layer 5 | layer 4 | layer 3 | layer 2 | ...
    side stream                          [pack 5-4][RCCL all-reduce 5-4][unpack]   [pack 3-2][all-reduce 3-2] ...

* reduce = "bf16" (default on the GPU): a bucket is packed as bf16(g / world) by `ug_grad_pack_bf16`, summed by RCCL
  over xGMI, and unpacked in place -- half the bytes on the links (3.1 GB instead of 6.2 GB per step for the 1.5B
  model) and half the time the collective's workgroups compete with the backward GEMMs for CUs.
* reduce = "fp32" (`UNIGEN_DDP_REDUCE=fp32`; always on CPU tensors): the bucket itself is all-reduced (AVG on RCCL, SUM and
  a scale on gloo) -- bit-for-bit what DDP's reducer computes.

After `finish()` every gradient holds the MEAN over ranks, exactly what DDP leaves in `.grad`: the caller's unchanged
`accelerator.clip_grad_norm_` and any stock torch optimizer see the same values as in the reference (no grad_scale
argument is needed any more; `grad_scale` stays as a constant 1.0 for older callers).

The engine installs this object by itself on the first backward of a process whose torch.distributed world is larger
than one (unigen_hip/modules.py: TrainEngine._dp_sync) and finishes it from an end-of-backward callback of the autograd
engine, so neither training/train.py nor bench.py calls anything here."""
import os

import torch
import torch.distributed as dist


class FlatGradSync:
    """engine: anything exposing .fp.grad (flat tensor), .fp.off (key -> (offset, shape)), .dims.num_hidden_layers and a
    settable .grad_ready_hook.  extra_params: callable returning the ordinary (non-flat) parameters whose gradients this
    object must average too (mm_projector when the model is not wrapped in DistributedDataParallel)."""

    def __init__(self, engine, process_group=None, layers_per_bucket=2, reduce=None, extra_params=None):
        self.engine, self.pg = engine, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.layers_per_bucket = max(1, int(layers_per_bucket))
        self.cuda = engine.fp.grad.is_cuda
        if reduce is None:
            reduce = os.environ.get("UNIGEN_DDP_REDUCE", "bf16" if self.cuda else "fp32")
        if reduce not in ("bf16", "fp32"):
            raise ValueError(f"FlatGradSync: reduce must be 'bf16' or 'fp32' (got {reduce!r})")
        if reduce == "bf16" and not self.cuda:
            raise ValueError("FlatGradSync: the bf16 exchange packs with a HIP kernel; CPU tensors use reduce='fp32'")
        self.reduce = reduce
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.extra_params = extra_params
        self.enabled = True
        self._pending = []              # CPU path: (work, view) still in flight
        self._hi = None                 # end (exclusive) of the not-yet-flushed region
        fp = engine.fp
        n = engine.dims.num_hidden_layers
        self._starts = {i: fp.off[f"l{i}.wqkv"][0] for i in range(n)}
        self._numel = fp.grad.numel()
        self._stage = None              # bf16 staging buffer, sized for the largest bucket on first use
        self.bytes_on_wire = 0          # payload handed to the collective since construction (tests / bench reporting)
        engine.grad_ready_hook = self.on_ready

    # ------------------------------------------------------------------ one bucket
    def _stage_for(self, n):
        if self._stage is None or self._stage.numel() < n:
            cap = max(n, self._starts.get(0, n))           # the embedding table (everything before layer 0) is the largest bucket
            self._stage = torch.empty(cap, dtype=torch.bfloat16, device=self.engine.fp.grad.device)
        return self._stage[:n]

    def _allreduce_mean_(self, buf):
        """in-place mean over ranks of an fp32 tensor on the current stream / host"""
        if self.backend == "nccl":
            dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            buf.mul_(1.0 / self.world)

    def _flush(self, lo, hi):
        if self.world == 1 or hi <= lo or not self.enabled:
            return
        buf = self.engine.fp.grad[lo:hi]
        if not self.cuda:
            self.bytes_on_wire += buf.numel() * 4
            self._pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), buf))
            return
        from . import ops
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            if self.reduce == "bf16":
                stage = self._stage_for(hi - lo)
                ops.grad_pack_bf16(buf, stage, 1.0 / self.world)
                dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.pg)
                ops.grad_unpack_bf16(stage, buf)
                self.bytes_on_wire += stage.numel() * 2
            else:
                self._allreduce_mean_(buf)
                self.bytes_on_wire += buf.numel() * 4

    # ------------------------------------------------------------------ driven by backward
    def begin(self, enabled=True):
        """Start of a backward pass.  enabled=False: a gradient-accumulation micro-step (DDP's no_sync): nothing is
        exchanged, gradients keep accumulating locally."""
        self.enabled = bool(enabled)
        self._hi = None

    def on_ready(self, tag):
        """tag: 'norm' (first), layer index N-1 .. 0, then 'embed' (last)."""
        if not self.enabled:
            return
        if tag == "norm":
            self._hi = self._numel
            return
        if tag == "embed":
            self._flush(0, self._hi if self._hi is not None else self._numel)
            self._hi = 0
            return
        i = int(tag)
        if i % self.layers_per_bucket == 0:
            lo = self._starts[i]
            self._flush(lo, self._hi if self._hi is not None else self._numel)
            self._hi = lo

    def finish(self):
        """End of backward: flush what no hook has covered (the embedding table when the caller passed embeddings, or
        everything if no hook fired), average the ordinary parameters' gradients, and make the current stream (the host,
        on CPU) wait for every outstanding bucket."""
        if not self.enabled or self.world == 1:
            self._hi = None
            return
        if self._hi is None:
            self._hi = self._numel
        if self._hi > 0:
            self._flush(0, self._hi)
        self._hi = None
        extra = [p for p in (self.extra_params() if self.extra_params is not None else []) if p.grad is not None]
        if self.cuda:
            if extra:
                with torch.cuda.stream(self.stream):
                    self.stream.wait_stream(torch.cuda.current_stream())
                    self._average_extra(extra)
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w, buf in self._pending:
                w.wait()
                buf.mul_(1.0 / self.world)
            self._pending = []
            if extra:
                self._average_extra(extra)

    def _average_extra(self, params):
        flat = torch.cat([p.grad.reshape(-1).float() for p in params])
        self._allreduce_mean_(flat)
        self.bytes_on_wire += flat.numel() * 4
        o = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n

    @property
    def grad_scale(self):
        return 1.0
