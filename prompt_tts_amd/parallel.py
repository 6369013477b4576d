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


def attach(model, process_group=None, bucket_bytes=64 << 20):
    """Create a reducer over model.store.flat_g and hook it into the model's backward."""
    st = model.store
    r = GradReducer(st.flat_g, lambda m: st.span(list(m.parameters())) if any(True for _ in m.parameters()) else (0, 0),
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
