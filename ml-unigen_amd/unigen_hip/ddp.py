"""Data-parallel gradient exchange for the flat gradient buffer (replaces the DDP / DeepSpeed reducer
behind `accelerator.backward`, reference training/train.py:775).

One process per GPU; the only collective on the training path is a SUM all-reduce of the flat fp32
gradient buffer, issued bucket by bucket on a side stream as backward retires layers (RCCL over
xGMI when the backend is "nccl"; gloo in the CPU tests), then the optimizer divides by world size
(`FusedAdamW.step(grad_scale=1/world)`) -- i.e. DDP's gradient averaging without a per-parameter
reducer or bucket copies."""
import torch
import torch.distributed as dist


class FlatGradSync:
    """engine: anything exposing .fp.grad (flat tensor), .fp.off (key -> (offset, shape)),
    .dims.num_hidden_layers and a settable .grad_ready_hook."""

    def __init__(self, engine, process_group=None, layers_per_bucket=4):
        self.engine, self.pg = engine, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.layers_per_bucket = layers_per_bucket
        self.cuda = engine.fp.grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self._pending = []
        self._hi = None                 # end (exclusive) of the not-yet-flushed region
        fp = engine.fp
        self._starts = {}
        n = engine.dims.num_hidden_layers
        for i in range(n):
            self._starts[i] = fp.off[f"l{i}.wqkv"][0]
        self._norm_start = fp.off["norm"][0]
        self._numel = fp.grad.numel()
        engine.grad_ready_hook = self.on_ready

    def _flush(self, lo, hi):
        if self.world == 1 or hi <= lo:
            return
        buf = self.engine.fp.grad[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            self._pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def on_ready(self, tag):
        """tag: 'norm' (first), layer index N-1 .. 0, then 'embed' (last)."""
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
            self._flush(lo, self._hi)
            self._hi = lo

    def finish(self):
        """Make the current stream (or the host, on CPU) wait for every outstanding bucket; if the
        embedding hook never fired (inputs_embeds path) the remaining head is flushed here."""
        if self._hi is None:
            self._hi = self._numel
        if self._hi > 0:
            self._flush(0, self._hi)
        self._hi = None
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w in self._pending:
                w.wait()
            self._pending = []

    @property
    def grad_scale(self):
        return 1.0 / self.world
