"""Data-parallel gradient reduction over RCCL/xGMI for the flat grad buffer (replaces torch DDP for this path).

The reference trains under accelerate -> torch DDP with find_unused_parameters=True (train.py:25-29,67-69),
i.e. 25 MB buckets, an extra "used-parameter" bitmap all-reduce per step and per-parameter autograd hooks.
Here the model's backward calls `on_ready(module)` when a sub-module's weight gradients are complete; since all
gradients live in ONE flat f32 buffer in registration order, a module is a contiguous slice and a bucket is a few
large slices.  Each bucket is all-reduced on a dedicated HIP stream as soon as it is ready, overlapping the rest
of backward; the unused `proj_out` parameters are simply zeros inside the slices (no bitmap, no graph walk).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, large collectives are what keeps every link busy.

Averaging is folded into the loss gradient (grad_scale = 1/world_size), so the collective is a plain SUM.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_grad, span_of, process_group=None, bucket_bytes=64 << 20):
        """flat_grad: 1-D f32 tensor; span_of(module) -> (lo, hi) element range of that module's parameters."""
        self.flat, self.span_of, self.pg = flat_grad, span_of, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self._pending, self._pending_bytes, self._done, self._work = [], 0, [], []
        self.launched = []          # [(lo, hi)] in launch order (inspected by tests)
        # called on the communication stream in front of every bucket: attach() makes that stream (not the compute stream)
        # wait for the weight-gradient side stream and fold the replicated small gradients, so backward never stalls
        self.pre_flush = None
        self.joins_side_stream = False
        # timing=True (bench.py, one instrumented step): HIP events on the COMMUNICATION stream around every bucket and one on
        # the compute stream where backward ends; timing_ms() then says how long the collectives ran and how much of that the
        # compute stream had to wait for
        self.timing = False
        self._ev = []
        self._ev_bwd_end = None

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def begin(self):
        self._pending, self._pending_bytes, self._done, self._work, self.launched = [], 0, [], [], []
        self._ev, self._ev_bwd_end = [], None

    def timing_ms(self):
        """After a step run with timing=True and a device synchronize: {"allreduce_ms": time the comm stream spent inside the
        buckets' collectives, "exposed_ms": how long after the end of backward the last collective finished (what the optimizer
        waits for), "overlap_pct": share of the collective time hidden under backward, "buckets", "bytes"}."""
        if not self._ev:
            return None
        busy = sum(e0.elapsed_time(e1) for e0, e1, _ in self._ev)
        exposed = max(0.0, self._ev_bwd_end.elapsed_time(self._ev[-1][1])) if self._ev_bwd_end is not None else busy
        return {"allreduce_ms": busy, "exposed_ms": min(exposed, busy), "buckets": len(self._ev),
                "overlap_pct": 100.0 * (1.0 - min(exposed, busy) / busy) if busy > 0 else 0.0,
                "bytes": sum(b for _, _, b in self._ev)}

    def on_ready(self, module):
        if self.world == 1:
            return
        lo, hi = self.span_of(module)
        if hi <= lo:
            return
        self._pending.append((lo, hi)); self._pending_bytes += 4 * (hi - lo)
        if self._pending_bytes >= self.bucket_bytes:
            self._flush()

    @staticmethod
    def _merge(ranges):
        out = []
        for lo, hi in sorted(ranges):
            if out and lo <= out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], hi))
            else:
                out.append((lo, hi))
        return out

    def _flush(self):
        if not self._pending:
            return
        ranges = self._merge(self._pending)
        self._pending, self._pending_bytes = [], 0
        if self.cuda:
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if self.pre_flush is not None:
                    self.pre_flush()
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(self.stream)
                for lo, hi in ranges:
                    dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
                if self.timing:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(self.stream)
                    self._ev.append((e0, e1, sum(4 * (hi - lo) for lo, hi in ranges)))
        else:
            for lo, hi in ranges:
                self._work.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._done += ranges; self.launched += ranges

    def finish(self):
        """Reduce whatever was not announced, then make the compute stream wait for every collective."""
        if self.world == 1:
            return
        covered = self._merge(self._done + self._pending)
        pos, rest = 0, []
        for lo, hi in covered:
            if lo > pos:
                rest.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.flat.numel():
            rest.append((pos, self.flat.numel()))
        self._pending += rest
        if self.cuda and self.timing:
            self._ev_bwd_end = torch.cuda.Event(enable_timing=True); self._ev_bwd_end.record(torch.cuda.current_stream())
        self._flush()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w in self._work:
                w.wait()


class ShardedGradReducer(GradReducer):
    """Reduce-scatter -> sharded clip + AdamW -> all-gather (MI355X-first form of the DDP + replicated-AdamW pair of
    train.py:41-47,67-69,115-120).

    xGMI is point-to-point, so an all-reduce IS a reduce-scatter followed by an all-gather; issuing the two halves ourselves
    puts the optimizer between them: every bucket (a contiguous range of the flat gradient buffer, announced by the backward
    like GradReducer's) is reduce-scattered in place on the communication stream while backward continues, so rank r ends up
    with the summed gradient of slice r of every bucket and updates ONLY those parameters (1/W of the 28-byte-per-parameter
    optimizer pass; Adam moments are only ever touched on their owner).  The updated values overwrite the gradient slice
    (pt_adamw_step_range, publish) and ONE in-place all-gather per bucket carries them to the other ranks, which adopt them into
    master + shadow (pt_import_params_range).  The global gradient norm is the sum of the ranks' slice norms (a 4-byte
    all-reduce).  Same bytes on the links as the all-reduce, one collective per bucket and direction.
    A bucket whose body is not a multiple of 4 W elements keeps a tail (< 4 W elements; and a head of < 4 if it does not
    start on a quad) that is all-reduced and updated redundantly on every rank.  The bucket geometry must be the same every step (the moments live on the owner): it is
    recorded on the first step and checked afterwards."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.rank = dist.get_rank(self.pg) if dist.is_initialized() else 0
        self.buckets = []           # this step's (head_lo, lo, shard, tail_lo, hi), see _split
        self._geometry = None

    def begin(self):
        super().begin()
        self.buckets = []

    def _split(self, lo, hi):
        """(head_lo, lo, shard, tail_lo, hi): body [lo, lo + W shard) with lo % 4 == 0 (the optimizer moves 16-byte quads), the
        head [head_lo, lo) and the tail [tail_lo, hi) are all-reduced.  ParamStore spans are 64-element aligned: no head there."""
        body_lo = min((lo + 3) // 4 * 4, hi)
        shard = (hi - body_lo) // (4 * self.world) * 4
        return lo, body_lo, shard, body_lo + self.world * shard, hi

    def _collect(self, ranges):
        for lo, hi in ranges:
            head_lo, lo, shard, tail_lo, hi = b = self._split(lo, hi)
            self.buckets.append(b)
            if shard:
                body = self.flat[lo:tail_lo]
                mine = body[self.rank * shard:(self.rank + 1) * shard]
                # the in-place contract of the collective (NCCL / RCCL): the output is exactly slice `rank` of the input
                assert mine.data_ptr() == body.data_ptr() + self.rank * shard * 4 and mine.numel() == shard
                w = dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=self.pg, async_op=not self.cuda)
                if not self.cuda:
                    self._work.append(w)
            for a, e in ((head_lo, lo), (tail_lo, hi)):
                if e > a:
                    w = dist.all_reduce(self.flat[a:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=not self.cuda)
                    if not self.cuda:
                        self._work.append(w)

    def _flush(self):
        if not self._pending:
            return
        ranges = self._merge(self._pending)
        self._pending, self._pending_bytes = [], 0
        if self.cuda:
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if self.pre_flush is not None:
                    self.pre_flush()
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(self.stream)
                self._collect(ranges)
                if self.timing:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(self.stream)
                    self._ev.append((e0, e1, sum(4 * (hi - lo) for lo, hi in ranges)))
        else:
            self._collect(ranges)
        self._done += ranges; self.launched += ranges

    def finish(self):
        super().finish()
        geo = tuple(self.buckets)
        if self._geometry is None:
            self._geometry = geo
        elif geo != self._geometry:
            raise RuntimeError("sharded optimizer: the gradient buckets changed between steps (the Adam moments live on "
                               "the rank that owned them before)")

    # -- what the optimizer needs -------------------------------------------------------------------------------------
    def owned_ranges(self):
        """[(lo, hi)] this rank updates and publishes, merged."""
        return self._merge([(lo + self.rank * sh, lo + (self.rank + 1) * sh) for _, lo, sh, _, _ in self.buckets if sh])

    def tail_ranges(self):
        """Heads and tails: all-reduced, updated redundantly by every rank (a few elements per bucket)."""
        return [(a, e) for h, lo, _, t, hi in self.buckets for a, e in ((h, lo), (t, hi)) if e > a]

    def foreign_ranges(self):
        out = []
        for _, lo, sh, tail_lo, _ in self.buckets:
            if sh and self.rank > 0:
                out.append((lo, lo + self.rank * sh))
            if sh and self.rank + 1 < self.world:
                out.append((lo + (self.rank + 1) * sh, tail_lo))
        return self._merge(out)

    def allgather_published(self, buf=None):
        """In-place all-gather of every bucket body of `buf` (default: the gradient buffer, which holds the published
        parameters after the sharded update; also used for the Adam moments when a checkpoint needs them whole)."""
        buf = self.flat if buf is None else buf
        work = []
        for _, lo, sh, tail_lo, _ in self.buckets:
            if not sh:
                continue
            body = buf[lo:tail_lo]
            mine = body[self.rank * sh:(self.rank + 1) * sh]
            src = mine if self.cuda else mine.clone()            # the CPU backends do not take an aliased input
            w = dist.all_gather_into_tensor(body, src, group=self.pg, async_op=not self.cuda)
            if not self.cuda:
                work.append(w)
        for w in work:
            w.wait()

    def global_sum(self, t):
        """Sum of a (1,) tensor over the ranks (the squared gradient norm of the slices)."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t


def attach(model, process_group=None, bucket_bytes=64 << 20, sharded=None):
    """Create a reducer over model.store.flat_g and hook it into the model's backward.  sharded (default: PT_DP_SHARDED=1):
    reduce-scatter + sharded optimizer + all-gather (ShardedGradReducer) instead of the all-reduce + replicated optimizer."""
    st = model.store
    if sharded is None:
        sharded = __import__("os").environ.get("PT_DP_SHARDED", "0") == "1"
    cls = ShardedGradReducer if sharded else GradReducer
    r = cls(st.flat_g, lambda m: st.span(list(m.parameters())) if any(True for _ in m.parameters()) else (0, 0),
            process_group, bucket_bytes)
    if r.cuda and __import__("os").environ.get("PT_DP_HOOK_JOIN", "0") != "1":
        from . import engine as E

        def pre_flush():                 # runs with the communication stream current
            E.join_side_stream(st.device)
            E.fold_grad_replicas(st.flat_g)
        r.pre_flush = pre_flush
        r.joins_side_stream = True
    model.grad_ready_hook = r.on_ready
    return r
