"""Data-parallel gradient exchange for the flat gradient buffer (replaces the DistributedDataParallel reducer that
`accelerator.prepare` / `accelerator.backward` put behind the reference's step, training/train.py:492,775;
SURVEY.md §2.2, §8e).

One process per GPU, full replica.  The only collective on the training path is the MEAN all-reduce of the flat
gradient buffer, issued bucket by bucket on a side HIP stream while backward is still retiring earlier layers:

    backward (main stream)   ... layer 5 | layer 4 | layer 3 | layer 2 | ...
    side stream                          [pack 5-4][RCCL all-reduce 5-4][unpack]   [pack 3-2][all-reduce 3-2] ...

* reduce = "fp32" (default): the bucket itself is all-reduced (AVG on RCCL, SUM and a scale on gloo) -- bit-for-bit
  what DDP's reducer computes; 6.2 GB per step on the links for the 1.5B model.
* reduce = "fp32_rsag" (`UNIGEN_DDP_REDUCE=fp32_rsag`): the same fp32 mean as ONE reduce-scatter + ONE all-gather in place on
  the bucket (every rank owns a slice of the sum; a short all-reduce covers the tail that does not divide by the world size).
* reduce = "bf16_fp32acc" (`UNIGEN_DDP_REDUCE=bf16_fp32acc`): bf16 on the wire, fp32 arithmetic.  Each rank packs
  bf16(g), an all-to-all hands rank r every rank's copy of slice r, the slice is summed in fp32 in rank order and scaled
  (`ug_grad_sum_shards_bf16`), rounded to bf16 ONCE and all-gathered: two bf16 roundings per element whatever the world
  size (worst element 2^-8 relative), 3.1 GB on the links.
* reduce = "bf16" (`UNIGEN_DDP_REDUCE=bf16`): bf16(g / world) summed by the collective itself in bf16 -- the cheapest
  form and the least accurate one: a ring adds world - 1 bf16 roundings of partial sums on top of the packing
  (tests/test_ddp_cpu.py::test_bf16_exchange_error_by_world_size measures every mode at 2 / 4 / 8 ranks).

Transport: `torch.distributed` collectives by default (backend "nccl" IS RCCL on ROCm); `UNIGEN_DDP_TRANSPORT=ug_comm` moves
the buckets through the library's own RCCL entry points instead (include/unigen_hip.h: ug_comm_*; csrc/comm.hip) -- the same
wire formats on the communicator's own side stream, torch.distributed only carries the 128-byte rendezvous id.

The tied embedding table (983 MB of fp32 for the 1.5B model, a sixth of the payload) has two kinds of writers: the head's DENSE
weight gradient, final right after the head's backward -- the very first thing a backward pass computes -- and the per-token
scatter-adds of the embedding lookups, which come last.  The dense part is handed over as soon as the last recorded head has run
(tag 'head', before 'norm') and travels under the whole decoder-stack backward; the lookups never touch the table before the
exchange: their (token id, gradient row) pairs -- 12 336 rows of 6 KB per rank for the benchmark batch, 76 MB -- are kept aside,
all-gathered at the end (padded to the largest per-rank count, agreed by a one-element MAX collective issued at the START of the
backward pass on the side stream, so the host never waits for the device), sorted by id with a stable sort and added by a
one-writer-per-row kernel (`ug_embed_bwd_sorted`): mean(head) + (1 / W) sum of every rank's lookup rows, the same bits on every
rank.  Any other order of writers stays correct: a dense writer that arrives after the hand-over waits for the exchange in
flight and marks the table for a second exchange at the end (the mean of identical values).  `UNIGEN_DDP_SPARSE_EMBED=0`
restores the single end-of-backward exchange of the whole table.

After `finish()` every gradient holds the MEAN over ranks, exactly what DDP leaves in `.grad`: the caller's unchanged
`accelerator.clip_grad_norm_` and any stock torch optimizer see the same values as in the reference (no grad_scale
argument is needed any more; `grad_scale` stays as a constant 1.0 for older callers).

The engine installs this object by itself on the first backward of a process whose torch.distributed world is larger
than one (unigen_hip/modules.py: TrainEngine._dp_sync) and finishes it from an end-of-backward callback of the autograd
engine, so neither training/train.py nor bench.py calls anything here."""
import os

import torch
import torch.distributed as dist

from .lib import UniGenHipError


class FlatGradSync:
    """engine: anything exposing .fp.grad (flat tensor), .fp.off (key -> (offset, shape)), .dims.num_hidden_layers and a
    settable .grad_ready_hook.  extra_params: callable returning the ordinary (non-flat) parameters whose gradients this
    object must average too (mm_projector when the model is not wrapped in DistributedDataParallel)."""

    def __init__(self, engine, process_group=None, layers_per_bucket=2, reduce=None, extra_params=None):
        self.engine, self.pg = engine, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # UNIGEN_DDP_FORCE=1: run the exchange at world size 1 too (the mean of one rank is the gradient itself) -- lets a
        # one-GPU box drive pack -> RCCL -> unpack on the side stream (tests/test_ddp_gpu.py)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("UNIGEN_DDP_FORCE", "0") == "1")
        self.layers_per_bucket = max(1, int(layers_per_bucket))
        self.cuda = engine.fp.grad.is_cuda
        if reduce is None:
            reduce = os.environ.get("UNIGEN_DDP_REDUCE", "fp32")
        if reduce not in ("bf16", "bf16_fp32acc", "fp32", "fp32_rsag"):
            raise ValueError(f"FlatGradSync: reduce must be 'fp32', 'fp32_rsag', 'bf16_fp32acc' or 'bf16' (got {reduce!r})")
        if reduce.startswith("bf16") and not self.cuda:
            raise ValueError("FlatGradSync: the bf16 exchanges pack with a HIP kernel; CPU tensors use reduce='fp32'")
        self.reduce = reduce
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # largest bucket handed to one collective, in elements (the tied embedding, 245.6 M elements for the 1.5B model, is
        # otherwise ONE exposed collective at the end of backward): UNIGEN_DDP_MAX_BUCKET_MB of fp32
        self.max_bucket = max(1 << 16, int(float(os.environ.get("UNIGEN_DDP_MAX_BUCKET_MB", "512")) * (1 << 20) / 4))
        self._overlap = True
        self.transport = os.environ.get("UNIGEN_DDP_TRANSPORT", "torch")
        if self.transport not in ("torch", "ug_comm"):
            raise ValueError(f"FlatGradSync: UNIGEN_DDP_TRANSPORT must be 'torch' or 'ug_comm' (got {self.transport!r})")
        self._comm = None
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
        self.early_embed_handovers = 0  # passes in which the tied table's dense part left right after the head's backward
        self.lookup_bytes_on_wire = 0   # ... of which the all-gathered embedding-lookup rows and ids (counted separately)
        # tied embedding table: dense head gradient handed over early, lookups exchanged as (id, row) pairs at the end
        emb = fp.off.get("embed")
        self.sparse_embed = (os.environ.get("UNIGEN_DDP_SPARSE_EMBED", "1") == "1" and emb is not None and len(emb[1]) == 2
                             and emb[0] == 0 and emb[1][1] % 4 == 0)
        self._embed_end = self._starts[0] if emb is not None and emb[0] == 0 else 0
        self._embed_done = False        # the table's dense part has been handed over in this pass
        self._lookups = []              # (ids int64 [n], rows fp32 [n, H]) kept aside in this pass
        self._cap = None                # agreed per-rank row capacity of this pass: int, or (event, pinned tensor) until read
        self.record_timeline = False    # bench: keep (bytes, issue / start / end events) of every bucket of the running pass
        self.timeline = []              # [(what, bytes, ev_issue on the compute stream, ev_start, ev_end on the exchange stream)]
        self._exposed = None            # (ev before, ev after) the compute stream's wait for the exchange at the end of backward
        self._pin = None
        engine.grad_ready_hook = self.on_ready
        if self.transport == "ug_comm":
            if not self.cuda:
                raise ValueError("FlatGradSync: the ug_comm transport (RCCL) moves device buffers only")
            self._comm_init()

    # ------------------------------------------------------------------ the library's own RCCL transport
    def _comm_init(self):
        import ctypes
        from . import lib as _l
        L = _l.load()
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _l.check(L.ug_comm_unique_id(ident), "ug_comm_unique_id")
        box = [bytes(ident.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=self.pg)           # the only use of torch.distributed on this transport
        cap = min(self.max_bucket, self._numel)
        handle = ctypes.c_void_p()
        _l.check(L.ug_comm_init(ctypes.byref(handle), self.world, self.rank, ctypes.create_string_buffer(box[0], 128), int(cap)), "ug_comm_init")
        self._comm, self._lib = handle, L
        self.backend = "ug_comm(rccl)"
        self._mode = {"fp32": 0, "bf16_fp32acc": 1, "bf16": 2, "fp32_rsag": 3}[self.reduce]

    def _comm_bucket(self, buf, mode=None):
        from . import lib as _l
        _l.check(self._lib.ug_comm_allreduce_bucket(self._comm, buf.data_ptr(), buf.numel(), self._mode if mode is None else mode,
                                                    torch.cuda.current_stream().cuda_stream), "ug_comm_allreduce_bucket")
        self.bytes_on_wire = int(self._lib.ug_comm_bytes_on_wire(self._comm))

    def __del__(self):
        if getattr(self, "_comm", None) is not None:
            try:
                self._lib.ug_comm_destroy(self._comm)
            except Exception:
                pass
            self._comm = None

    # ------------------------------------------------------------------ one bucket
    def _stage_for(self, n):
        """bf16 staging: [send n_pad | recv n_pad] for the largest bucket this object will ever hand over"""
        if self._stage is None or self._stage.numel() < 2 * n:
            cap = max(n, min(self.max_bucket, self._starts.get(0, n)) + 8 * self.world)
            self._stage = torch.empty(2 * cap, dtype=torch.bfloat16, device=self.engine.fp.grad.device)
        return self._stage

    def _coll(self, fn, out, inp, **kw):
        """One tensor-API collective (reduce_scatter_tensor / all_gather_into_tensor / all_to_all_single) on the exchange's
        process group.  RCCL takes device tensors; a rehearsal transport (gloo) is handed host copies, so the slice arithmetic
        around the call is the one an RCCL run executes (the rank-dependent paths must run with world > 1)."""
        if self.backend == "nccl" or not out.is_cuda:
            fn(out, inp, group=self.pg, **kw)
        else:
            o, i = out.cpu(), inp.cpu()
            fn(o, i, group=self.pg, **kw)
            out.copy_(o)

    def _allreduce_mean_(self, buf):
        """in-place mean over ranks of an fp32 tensor on the current stream / host"""
        if self.backend == "nccl":
            dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            buf.mul_(1.0 / self.world)

    def _reduce_scatter_all_gather_mean_(self, buf):
        """fp32 mean as reduce-scatter + all-gather in place (reduce = 'fp32_rsag'); the tail that does not divide by the world
        size goes through a short all-reduce"""
        W = self.world
        part = buf.numel() // (W * 4) * 4
        body = part * W
        if part:
            mine = buf[self.rank * part:(self.rank + 1) * part]
            if self.backend == "nccl":
                dist.reduce_scatter_tensor(mine, buf[:body], op=dist.ReduceOp.AVG, group=self.pg)
            else:
                self._coll(dist.reduce_scatter_tensor, mine, buf[:body], op=dist.ReduceOp.SUM)
                mine.mul_(1.0 / W)
            self._coll(dist.all_gather_into_tensor, buf[:body], mine)
        if buf.numel() > body:
            self._allreduce_mean_(buf[body:])

    def _exchange_bf16_fp32acc(self, buf):
        """bf16 on the wire, fp32 arithmetic (see the module docstring); runs on the current (side) stream"""
        from . import ops
        n, W = buf.numel(), self.world
        chunk = -(-n // (8 * W)) * 8                 # slice length per rank, 16-byte rows
        n_pad = chunk * W
        stage = self._stage_for(n_pad)
        send, recv = stage[:n_pad], stage[n_pad:2 * n_pad]
        if n_pad > n:
            send[n:].zero_()
        ops.grad_pack_bf16(buf, send[:n], 1.0)
        self._coll(dist.all_to_all_single, recv, send)                    # recv[j] = rank j's copy of my slice
        mine = send[self.rank * chunk:(self.rank + 1) * chunk]            # (my own packed slice is no longer needed)
        ops.grad_sum_shards_bf16(recv, W, chunk, mine, 1.0 / W)
        self._coll(dist.all_gather_into_tensor, recv, mine)
        ops.grad_unpack_bf16(recv[:n], buf)
        self.bytes_on_wire += 2 * n_pad * 2

    def _flush(self, lo, hi):
        if not self.active or hi <= lo or not self.enabled:
            return
        for a in range(lo, hi, self.max_bucket):                            # (one piece unless the span is the embedding table)
            self._flush_piece(a, min(hi, a + self.max_bucket))

    def _flush_piece(self, lo, hi):
        buf = self.engine.fp.grad[lo:hi]
        if not self.cuda:
            self.bytes_on_wire += buf.numel() * 4
            if self.reduce == "fp32_rsag":
                self._reduce_scatter_all_gather_mean_(buf)
                return
            self._pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), buf))
            return
        if self._comm is not None:
            self._comm_bucket(buf)
            return
        from . import ops
        rec = self.record_timeline
        ev = torch.cuda.Event(enable_timing=rec)
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            if rec:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.timeline.append((f"[{lo}, {hi})", (hi - lo) * 4, ev, e0, e1))
            if self.reduce == "bf16":
                stage = self._stage_for(hi - lo)[:hi - lo]
                ops.grad_pack_bf16(buf, stage, 1.0 / self.world)
                dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.pg)
                ops.grad_unpack_bf16(stage, buf)
                self.bytes_on_wire += stage.numel() * 2
            elif self.reduce == "bf16_fp32acc":
                self._exchange_bf16_fp32acc(buf)
            elif self.reduce == "fp32_rsag":
                self._reduce_scatter_all_gather_mean_(buf)
                self.bytes_on_wire += buf.numel() * 4
            else:
                self._allreduce_mean_(buf)
                self.bytes_on_wire += buf.numel() * 4
            if rec:
                e1.record()

    # ------------------------------------------------------------------ embedding lookups kept aside
    def wants_lookups(self):
        """True while a backward pass that exchanges gradients keeps embedding-lookup gradients aside as (id, row) pairs."""
        return self.active and self.enabled and self.sparse_embed

    def _agree_capacity(self, rows_live, heads_live=0):
        """Start of a backward pass: every rank announces (a) its count of lookup rows recorded on its autograd graphs (known on the
        HOST since the forward) -> the MAX over ranks = the padded length of the end-of-backward all-gather, and (b) its count of
        recorded head segments (dense writers of the tied table) -> the early hand-over of the table right after the heads'
        backward is used in this pass ONLY if every rank announces the same, non-zero count.  Both decisions change the
        sequence of collectives a rank issues, so neither may be taken from local state alone (ranks that diverge
        -- one falls back to the dense lookup gradient, a head is outside the loss on one rank -- would pair different
        collectives: a hang, not an error).  On a GPU transport the collective and the copy back to pinned memory run on the side
        stream, which is idle now: the result is on the host long before anything asks for it."""
        n, h = int(rows_live), int(heads_live)
        if not self.cuda or self.backend not in ("nccl", "ug_comm(rccl)"):
            if self.world == 1:
                self._cap = (n, h > 0)
                return
            mine = torch.tensor([n, h], dtype=torch.int64)
            every = [torch.zeros(2, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(every, mine, group=self.pg)
            every = torch.stack(every)
            self._cap = (int(every[:, 0].max()), bool((every[:, 1] == every[0, 1]).all()) and int(every[0, 1]) > 0)
        else:                                     # (also at world size 1 with the exchange forced on: the one-GPU boxes drive this path)
            dev = self.engine.fp.grad.device
            if self._pin is None:
                self._pin = (torch.zeros(2, dtype=torch.int64).pin_memory(), torch.zeros(2 * self.world, dtype=torch.int64).pin_memory(),
                             torch.zeros(2, dtype=torch.int64, device=dev), torch.zeros(2 * self.world, dtype=torch.int64, device=dev))
            src, dst, d_src, d_all = self._pin
            src[0], src[1] = n, h
            with torch.cuda.stream(self.stream):
                d_src.copy_(src, non_blocking=True)
                if self._comm is not None:
                    from . import lib as _l
                    side = torch.cuda.current_stream().cuda_stream
                    _l.check(self._lib.ug_comm_allgather(self._comm, d_src.data_ptr(), d_all.data_ptr(), 16, side), "ug_comm_allgather")
                    _l.check(self._lib.ug_comm_wait(self._comm, side), "ug_comm_wait")
                else:
                    dist.all_gather_into_tensor(d_all, d_src, group=self.pg)
                dst.copy_(d_all, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._cap = (ev, dst)

    def _plan(self, block=True):
        """(agreed row capacity, early hand-over agreed) of the running pass.  block=False: None while the ranks' answer has not
        reached the host yet (no host wait)."""
        if self._cap is None:
            return 0, False
        if not isinstance(self._cap[0], int):
            ev, dst = self._cap
            if not block and not ev.query():
                return None
            ev.synchronize()
            every = dst.view(self.world, 2)
            self._cap = (int(every[:, 0].max()), bool((every[:, 1] == every[0, 1]).all()) and int(every[0, 1]) > 0)
        return self._cap

    def _capacity(self):
        return self._plan()[0]

    def early_handover_agreed(self, block=True):
        plan = self._plan(block)
        return None if plan is None else plan[1]

    def set_reduce(self, reduce):
        """Switch the exchange's arithmetic between passes (bench.py times the default and `bf16_fp32acc` in one process).  Every rank
        must switch at the same point: the modes issue different collectives."""
        if reduce not in ("bf16", "bf16_fp32acc", "fp32", "fp32_rsag"):
            raise ValueError(f"FlatGradSync: reduce must be 'fp32', 'fp32_rsag', 'bf16_fp32acc' or 'bf16' (got {reduce!r})")
        if reduce.startswith("bf16") and not self.cuda:
            raise ValueError("FlatGradSync: the bf16 exchanges pack with a HIP kernel; CPU tensors use reduce='fp32'")
        self.reduce = reduce
        self._mode = {"fp32": 0, "bf16_fp32acc": 1, "bf16": 2, "fp32_rsag": 3}[reduce]

    def add_lookup(self, ids, rows):
        """An embedding lookup's backward: ids int64 [n], rows fp32 [n, H] (kept, not copied)."""
        self._lookups.append((ids.reshape(-1), rows))

    def before_dense_embed_write(self):
        """A dense writer of the tied table (a head's weight gradient, a lookup that could not be kept aside) is about to run.
        If the table has already been handed over in this pass, wait for that exchange and schedule another one at the end
        (what is there is the mean over ranks already: averaging identical values again changes nothing)."""
        if not self._embed_done:
            return
        if self.world > 1:
            # the other ranks will not exchange the table a second time: carrying on would pair different collectives (a hang)
            raise UniGenHipError("data-parallel exchange: a dense writer of the tied embedding table ran after the table's agreed early "
                                 "hand-over on this rank (an embedding lookup that was not recorded before backward started, or more "
                                 "lookup rows than announced); the ranks' collectives would no longer pair")
        if self.cuda:
            if self._comm is not None:
                from . import lib as _l
                _l.check(self._lib.ug_comm_wait(self._comm, torch.cuda.current_stream().cuda_stream), "ug_comm_wait")
            else:
                torch.cuda.current_stream().wait_stream(self.stream)
        else:
            self._drain_pending()
        self._embed_done = False

    def _drain_pending(self):
        for w, buf in self._pending:
            w.wait()
            buf.mul_(1.0 / self.world)
        self._pending = []

    def _all_gather_flat(self, recv, send):
        """recv [W * n] <- every rank's send [n] (on the current stream / host)"""
        if self._comm is not None:
            from . import lib as _l
            side = torch.cuda.current_stream().cuda_stream
            _l.check(self._lib.ug_comm_allgather(self._comm, send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size(), side),
                     "ug_comm_allgather")
            _l.check(self._lib.ug_comm_wait(self._comm, side), "ug_comm_wait")
        else:
            self._coll(dist.all_gather_into_tensor, recv, send)

    def _exchange_lookups(self):
        """End of backward, on the side stream (the host, on CPU): all-gather the padded (id, row) pairs of every rank and add
        (1 / W) x their sum to the table, deterministically (stable sort by id, one writer per table row)."""
        cap = self._capacity()
        lookups, self._lookups = self._lookups, []
        if cap == 0:
            return
        fp = self.engine.fp
        (o, (V, H)) = fp.off["embed"]
        table = fp.grad[o:o + V * H].view(V, H)
        n = sum(i.numel() for i, _ in lookups)
        dev = fp.grad.device
        ids = torch.full((cap,), -1, dtype=torch.int64, device=dev)
        rows = torch.empty((cap, H), dtype=torch.float32, device=dev)
        a = 0
        for i, r in lookups:
            ids[a:a + i.numel()] = i
            rows[a:a + i.numel()] = r.reshape(-1, H)
            a += i.numel()
        if n < cap:
            rows[n:].zero_()
        W = self.world
        if W > 1 or (self.cuda and self.backend in ("nccl", "ug_comm(rccl)")):       # (world 1 on RCCL: a one-rank gather, for coverage)
            ids_all = torch.empty(W * cap, dtype=torch.int64, device=dev)
            rows_all = torch.empty((W * cap, H), dtype=torch.float32, device=dev)
            self._all_gather_flat(ids_all, ids)
            self._all_gather_flat(rows_all.view(-1), rows.view(-1))
            self.lookup_bytes_on_wire += W * cap * (8 + 4 * H)
        else:
            ids_all, rows_all = ids, rows
        if self.cuda:
            from . import ops
            ids_sorted, order = torch.sort(ids_all, stable=True)
            ops.embed_bwd_sorted(ids_sorted, order, rows_all, table, 1.0 / W)
        else:
            keep = ids_all >= 0
            table.index_add_(0, ids_all[keep], rows_all[keep], alpha=1.0 / W)       # sequential on the host: deterministic

    # ------------------------------------------------------------------ driven by backward
    def begin(self, enabled=True, lookup_rows=0, heads_live=0):
        """Start of a backward pass.  enabled=False: a gradient-accumulation micro-step (DDP's no_sync): nothing is
        exchanged, gradients keep accumulating locally.  lookup_rows: embedding-lookup rows recorded on this rank's live
        autograd graphs (an upper bound of what this pass will keep aside); heads_live: recorded head segments (dense writers of
        the tied table) -- the ranks agree on both before anything depends on them (_agree_capacity)."""
        self.enabled = bool(enabled)
        self._hi = None
        self._overlap = True
        self._embed_done = False
        self._lookups = []
        self._cap = None
        self.timeline = []
        self._exposed = None
        if self.wants_lookups():
            self._agree_capacity(lookup_rows, heads_live)

    def set_overlap(self, on):
        """Called at the start of every decoder-stack segment of a backward pass: only the LAST segment that writes the
        layers' gradients may hand them over while it runs (a forward that ran the stack twice -- chosen / rejected, two
        micro-batches under one backward -- has an earlier segment whose hooks must be ignored)."""
        self._overlap = bool(on)

    def on_ready(self, tag):
        """tag: 'head' (the tied table's dense part is final), 'norm', layer index N-1 .. 0, then 'embed' (any number of times, last).
        The embedding table is NEVER handed over from here: a forward may look embeddings up several times (the reference's
        callers do: training/train.py:602-609,633,671 -- text, t2i and mmu parts), the tied head writes the same table, and
        only the end of backward (finish) knows that no writer is left."""
        if tag == "head":
            # the last recorded head has written its dense weight gradient into the tied table: nothing but lookups (kept
            # aside) is expected to touch it any more in this pass
            # (the ranks' agreement is READ without a host wait here: backward has only the head's kernels queued at this point, and a
            # rank that is ahead of the others would stall its launch queue for their all-gather -- ADVICE r5.  If the answer is not on
            # the host yet, the hand-over happens at the next tag through _ensure_early_done, at the same position in the sequence of
            # collectives, with the layers' backward already queued behind it.)
            if (self.enabled and self.active and self.sparse_embed and not self._embed_done and self._embed_end > 0
                    and self.early_handover_agreed(block=False)):
                self._flush(0, self._embed_end)
                self._embed_done = True
                self.early_embed_handovers += 1
            return
        if not self.enabled or not self._overlap:
            return
        self._ensure_early_done()
        if tag == "norm":
            self._hi = self._numel
            return
        if tag == "embed":
            return
        i = int(tag)
        if self._hi is not None and i % self.layers_per_bucket == 0:
            lo = self._starts[i]
            if lo < self._hi:
                self._flush(lo, self._hi)
                self._hi = lo

    def _ensure_early_done(self):
        """Every rank announced the same non-zero number of recorded heads, so every rank hands the table over as its FIRST
        collective of the pass.  Normally that happens at the 'head' tag.  If this rank reaches its first other collective without it
        -- a head segment that was recorded but is not part of this loss (its graph was dropped, or it belongs to another
        backward) keeps the count from reaching zero -- the hand-over is issued here, at the same position in the collective
        sequence: the heads run first in any backward, so what they wrote is in the table by now.  A dense writer that still
        shows up afterwards is refused (before_dense_embed_write), never re-exchanged from local state."""
        if (self.sparse_embed and self._embed_end > 0 and not self._embed_done and self._cap is not None
                and self.early_handover_agreed()):
            self._flush(0, self._embed_end)
            self._embed_done = True
            self.early_embed_handovers += 1

    def finish(self):
        """End of backward: flush what no hook has covered (always the embedding table; everything if no hook fired or
        overlapping was off), average the ordinary parameters' gradients, and make the current stream (the host, on CPU)
        wait for every outstanding bucket."""
        if not self.enabled or not self.active:
            self._hi = None
            return
        self._ensure_early_done()
        if self._hi is None:
            self._hi = self._numel
        lo = self._embed_end if self._embed_done else 0
        if self._hi > lo:
            self._flush(lo, self._hi)
        self._hi = None
        if self.sparse_embed:
            if not self.cuda:
                self._drain_pending()
                self._exchange_lookups()
            else:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    self.stream.wait_event(ev)
                    if self._comm is not None:      # the dense buckets run on the communicator's own stream
                        from . import lib as _l
                        _l.check(self._lib.ug_comm_wait(self._comm, self.stream.cuda_stream), "ug_comm_wait")
                    self._exchange_lookups()
        self._embed_done = False
        extra = [p for p in (self.extra_params() if self.extra_params is not None else []) if p.grad is not None]
        if self._comm is not None:
            from . import lib as _l
            flat = torch.cat([p.grad.reshape(-1).float() for p in extra]) if extra else None
            if flat is not None:
                self._comm_bucket(flat, mode=0)
            _l.check(self._lib.ug_comm_wait(self._comm, torch.cuda.current_stream().cuda_stream), "ug_comm_wait")
            torch.cuda.current_stream().wait_stream(self.stream)        # (the lookups' sort + scatter ran there)
            o = 0
            for p in extra:
                n = p.grad.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
        elif self.cuda:
            if extra:
                with torch.cuda.stream(self.stream):
                    self.stream.wait_stream(torch.cuda.current_stream())
                    self._average_extra(extra)
            if self.record_timeline:
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record(torch.cuda.current_stream())
            torch.cuda.current_stream().wait_stream(self.stream)
            if self.record_timeline:
                w1.record(torch.cuda.current_stream())
                self._exposed = (w0, w1)
        else:
            self._drain_pending()
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

    # ------------------------------------------------------------------ reporting
    def describe(self):
        return {"transport": self.transport, "backend": self.backend, "reduce": self.reduce, "world": self.world,
                "collective": {"fp32": "all_reduce(AVG)", "fp32_rsag": "reduce_scatter(AVG) + all_gather",
                               "bf16_fp32acc": "all_to_all + fp32 sum + all_gather", "bf16": "all_reduce(bf16 SUM)"}[self.reduce],
                "tied_embedding": "dense head gradient after the head's backward, lookups as (id, row) pairs at the end"
                                  if self.sparse_embed else "one dense exchange at the end of backward",
                "layers_per_bucket": self.layers_per_bucket}

    def timeline_report(self):
        """Per bucket of the LAST backward pass (record_timeline on, torch transport on a GPU; call after a device sync): bytes, how
        long after the first bucket's hand-over it was handed over, how long it waited for the exchange stream, how long the
        collective took; and the exposed tail -- the time the compute stream stood waiting for the exchange at the end of backward.
        One driver run at N > 1 then says where the exchange's time is."""
        if not self.timeline:
            return None
        first = self.timeline[0][2]
        rows = []
        for what, nbytes, ev_issue, e0, e1 in self.timeline:
            ms = e0.elapsed_time(e1)
            rows.append({"span": what, "mb": round(nbytes / 1e6, 1), "issued_ms": round(first.elapsed_time(ev_issue), 3),
                         "queued_ms": round(ev_issue.elapsed_time(e0), 3), "collective_ms": round(ms, 3),
                         "gb_per_s": round(nbytes / max(ms, 1e-6) / 1e6, 1)})
        out = {"buckets": rows, "collective_ms_total": round(sum(r["collective_ms"] for r in rows), 3)}
        if self._exposed is not None:
            out["exposed_wait_ms"] = round(self._exposed[0].elapsed_time(self._exposed[1]), 3)
        return out

    def ranks_seen(self):
        """Number of ranks that answer on the communicator the buckets move through: every rank marks its own slot of a
        64-slot vector, the vector goes through the SAME exchange as a gradient bucket (mean over ranks), and the non-zero slots
        are counted."""
        dev = self.engine.fp.grad.device
        v = torch.zeros(max(64, self.world), dtype=torch.float32, device=dev)
        v[self.rank] = 1.0
        if self._comm is not None:
            from . import lib as _l
            self._comm_bucket(v, mode=0)
            _l.check(self._lib.ug_comm_wait(self._comm, torch.cuda.current_stream().cuda_stream), "ug_comm_wait")
        elif dist.is_initialized():
            self._allreduce_mean_(v)
        return int((v > 0).sum().item())

    @property
    def grad_scale(self):
        return 1.0
