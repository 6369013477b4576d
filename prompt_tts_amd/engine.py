"""Host-side execution engine: flat parameter store + layer forward/backward built from the C-ABI kernels.

Layout decisions (MI355X-first):
  * every activation is token-major (B*N, C) in the activation dtype (bf16 for speed, f32 for parity);
    there is no (B,C,N)<->(B,N,C) flip and no channel-concat copy anywhere in the model;
  * ALL parameters live in one flat f32 master buffer (nn.Parameters are views into it, so state_dict
    keys/shapes are the reference's), with flat f32 grad / Adam m / v buffers beside it and one
    activation-dtype "shadow" copy in kernel layout (Conv1d k=3 weights as [Cout][3][Cin]);
  * weight gradients are written by the wgrad kernels straight into the flat grad buffer (f32 atomics,
    split-K) -- one fused AdamW launch, one norm reduction and a handful of large all-reduce buckets.
"""
import ctypes as C
import math

import torch
from torch import nn

from . import _lib as L
from . import ops

ALIGN = 64  # elements; keeps every tensor 16-byte aligned in both f32 and bf16 buffers


def _round_up(x, a):
    return (x + a - 1) // a * a


_STORES = __import__("weakref").WeakSet()
DGRAD_PRETRANSPOSED = __import__("os").environ.get("PT_DGRAD_PRETRANSPOSED", "1") != "0"


def transposed_weight(w):
    """The W^T copy some ParamStore keeps for the shadow view `w`, or None."""
    for st in _STORES:
        if st.device == w.device:
            t = st.wt(w)
            if t is not None:
                return t
    return None


def conv_dgrad_weight(w3, cout, cin_pad):
    """The flipped + transposed copy [cin_pad][3*cout] some ParamStore keeps for the conv k3 weight shadow `w3`, or None."""
    for st in _STORES:
        if st.device == w3.device:
            t = st.wd(w3, cout, cin_pad)
            if t is not None:
                return t
    return None


def geglu_interleave_ok(F2, d):
    """May the GEGLU projection weight [F2][d] keep its bf16 shadow rows interleaved (value / gate of a column in one lane's
    accumulators: GEGLU in the ff1 epilogue)?  Its weight gradient can then only be un-interleaved by the fold of the GROUPED
    weight-gradient launch, so the predicate is exactly what _queue_wgrad accepts for dW [F2][d]: both sides >= 128 and at most
    WGRAD_GROUP_WGS tiles of 256 x 256 -- d = 64 or d >= 1536 (288 tiles) train with the plain layout + stand-alone GEGLU."""
    return (GEGLU_FUSED and WGRAD_GROUPED and F2 % 512 == 0 and d % 8 == 0 and F2 >= 128 and d >= 128
            and math.ceil(F2 / 256) * math.ceil(d / 256) <= WGRAD_GROUP_WGS)


class ParamStore:
    """Flat storage for every parameter of `module` on `device`; `dtype` is the activation/shadow dtype."""

    def __init__(self, module, device, dtype):
        self.device, self.dtype, self.pt = device, dtype, ops._DT[dtype]
        self.names, self.params, self.info = [], [], {}
        self.geglu_ids = set()          # parameters whose shadow rows are GEGLU-interleaved
        off = soff = 0
        segs = []
        # Every ResnetBlock1D.time_emb_proj is packed, in forward order, into ONE contiguous "late" region at the end of
        # the flat buffers (weights, then biases), so the 2L+... per-resnet time-embedding projections of a step are a
        # single GEMM forward and two backward instead of dozens of M = B launches.
        named = list(module.named_parameters())
        self._model_order = [p for _, p in named]
        # Likewise the K / V projections of every UNet cross-attention layer (they all read the same text-encoder output) are
        # packed into one [sum 2C][d_ctx] matrix: one forward GEMM, one dgrad and one weight gradient per step instead of one per
        # layer (12 launches of M = B*S rows each, far too small to fill the chip).  "Late" tensors are excluded from block spans
        # (their gradients complete at the END of backward): the data-parallel reducer picks them up in finish().
        is_tp = lambda nm: nm.endswith("time_emb_proj.weight") or nm.endswith("time_emb_proj.bias")
        is_kv = lambda nm: KV_BATCHED and nm.startswith("unet.") and (nm.endswith(".attn2.to_k.weight") or nm.endswith(".attn2.to_v.weight"))
        is_late = lambda nm: is_tp(nm) or is_kv(nm)
        ordered = [(nm, p) for nm, p in named if not is_late(nm)] + \
                  [(nm, p) for nm, p in named if nm.endswith("time_emb_proj.weight")] + \
                  [(nm, p) for nm, p in named if nm.endswith("time_emb_proj.bias")] + \
                  [(nm, p) for nm, p in named if is_kv(nm)]
        self.late = {"w": [], "b": [], "kv": []}
        for name, p in ordered:
            n = p.numel()
            late = is_late(name)
            is_conv3 = p.dim() == 3 and p.shape[2] == 3
            frozen = name.endswith("proj_out.weight") or name.endswith("proj_out.bias")
            # GEGLU projection (diffusers FeedForward.net[0].proj, [2F][d]): in bf16 its shadow rows are interleaved so the ff1
            # GEMM epilogue can apply value * gelu(gate) in registers (csrc/gemm.hip act 2 / 3)
            geglu = (dtype == torch.bfloat16 and name.endswith("ff.net.0.proj.weight") and p.dim() == 2
                     and geglu_interleave_ok(p.shape[0], p.shape[1]))
            if geglu:
                self.geglu_ids.add(id(p))
            if is_conv3:
                cout, cin = p.shape[0], p.shape[1]
                cin_pad, cout_pad = _round_up(cin, 8), _round_up(cout, 8)
                sn = cout_pad * 3 * cin_pad
                seg = (off, n, soff, 1, cin, cin_pad, int(frozen))
                sshape = (cout_pad, 3 * cin_pad)
            else:
                sn = n
                cin = cin_pad = 0
                seg = (off, n, soff, 2 if geglu else 0, p.shape[1] if geglu else 0, 0, int(frozen))
                sshape = (p.shape[0], n // p.shape[0]) if p.dim() >= 2 else (n,)
            self.names.append(name); self.params.append(p)
            self.info[id(p)] = dict(off=off, n=n, soff=soff, sn=sn, sshape=sshape, frozen=frozen, name=name, late=late)
            segs.append(seg)
            if late:
                if n % 8 != 0:
                    raise ValueError("packed (late) tensors must be multiples of 8 elements")
                kind = "kv" if is_kv(name) else ("w" if name.endswith("weight") else "b")
                if kind == "kv" and not self.late["kv"]:  # the packed K/V matrix is a GEMM operand: start it on a 256-byte line
                    shift = _round_up(off, ALIGN) - off
                    shift_s = _round_up(soff, ALIGN) - soff
                    off += shift; soff += shift_s
                    seg = (off, n, soff) + seg[3:]
                    self.info[id(p)].update(off=off, soff=soff)
                    segs[-1] = seg
                self.late[kind].append(p)
                off += n; soff += n                       # packed back to back
            else:
                off += _round_up(n, ALIGN); soff += _round_up(sn, ALIGN)
        off = _round_up(off, ALIGN); soff = _round_up(soff, ALIGN)
        self.n_total, self.n_shadow, self.n_seg = off, soff, len(segs)
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(soff, dtype=dtype, device=device)
        self.adam_m = self.adam_v = None
        self.step_count = 0
        arr = (L.pt_param_seg * len(segs))()
        for i, s in enumerate(segs):
            arr[i].offset, arr[i].numel, arr[i].shadow_offset, arr[i].layout, arr[i].cin, arr[i].cin_pad, arr[i].frozen = s
        raw = bytes(arr)
        self.seg_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.arena = None               # ZeroArena, created on first fused step
        self.arena_active = None        # set by the fused loss+backward path only (forward and backward are one episode)
        self.shadow_dirty = True
        self.fp8 = False                # enable_fp8(): fp8 operands for the GEMMs fp8_pays() selects (bf16 shadows only)
        self.shadow_version = 0         # bumped whenever the shadow is rewritten: fp8 weight copies are re-quantised lazily
        self._w8 = {}
        self._wt, self._wt_table, self._wt_version = {}, None, -1     # transposed shadow copies for the data-gradient GEMMs
        _STORES.add(self)
        with torch.no_grad():
            for p in self.params:
                i = self.info[id(p)]
                view = self.flat_p[i["off"]:i["off"] + i["n"]].view(p.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = None
        self.attach_grads()
        self.refresh_shadow()

    def params_in_model_order(self):
        """Parameters in module.parameters() order (what torch.optim.AdamW(model.parameters()) indexes); self.params is in
        FLAT-BUFFER order, which moves the time_emb_proj tensors to the end."""
        return self._model_order

    # -- views -------------------------------------------------------------------------------------------
    def w(self, p):
        i = self.info[id(p)]
        return self.shadow[i["soff"]:i["soff"] + i["sn"]].view(i["sshape"])

    def g(self, p):
        """Gradient storage as the kernels write it: Conv1d k=3 weights as [Cout][3*Cin], everything else as p.shape."""
        i = self.info[id(p)]
        flat = self.flat_g[i["off"]:i["off"] + i["n"]]
        if p.dim() == 3 and p.shape[2] == 3:
            return flat.view(p.shape[0], 3 * p.shape[1])
        return flat.view(p.shape)

    def grad_view(self, p):
        """The gradient in the parameter's own (reference) shape -- a strided view for Conv1d k=3 weights."""
        g = self.g(p)
        if p.dim() == 3 and p.shape[2] == 3:
            return g.view(p.shape[0], 3, p.shape[1]).permute(0, 2, 1)
        return g

    def f(self, p):
        return p.data

    def fused(self, ps):
        """(shadow2d, grad2d) over adjacent 2-D parameters stacked along rows (e.g. to_q|to_k|to_v), or None."""
        infos = [self.info[id(p)] for p in ps]
        cols = ps[0].shape[1]
        for a, b, pa in zip(infos[:-1], infos[1:], ps[:-1]):
            if a["off"] + a["n"] != b["off"] or a["soff"] + a["sn"] != b["soff"] or pa.shape[1] != cols:
                return None
        rows = sum(p.shape[0] for p in ps)
        i0 = infos[0]
        return (self.shadow[i0["soff"]:i0["soff"] + rows * cols].view(rows, cols),
                self.flat_g[i0["off"]:i0["off"] + rows * cols].view(rows, cols))

    def span(self, ps):
        """[lo, hi) element range of the flat buffers covering parameters ps (registration-contiguous)."""
        infos = [self.info[id(p)] for p in ps if not self.info[id(p)]["late"]]
        if not infos:
            return 0, 0
        return min(i["off"] for i in infos), max(i["off"] + _round_up(i["n"], ALIGN) for i in infos)

    def late_kv_views(self):
        """(W [sum 2C][d_ctx] shadow, dW f32, {id(weight): first row}) over the packed cross-attention K/V projections, or None."""
        ps = self.late["kv"]
        if not ps:
            return None
        i0 = self.info[id(ps[0])]
        cols = ps[0].shape[1]
        rows, where = 0, {}
        for p in ps:
            if p.shape[1] != cols or self.info[id(p)]["off"] != i0["off"] + rows * cols:
                return None
            where[id(p)] = rows
            rows += p.shape[0]
        return (self.shadow[i0["soff"]:i0["soff"] + rows * cols].view(rows, cols),
                self.flat_g[i0["off"]:i0["off"] + rows * cols].view(rows, cols), where)

    def late_views(self):
        """(W [Ct][K] f32 master, dW, b [Ct], db) over all time_emb_proj tensors, or None when there are none."""
        if not self.late["w"]:
            return None
        iw, ib = self.info[id(self.late["w"][0])], self.info[id(self.late["b"][0])]
        K = self.late["w"][0].shape[1]
        Ct = sum(p.shape[0] for p in self.late["w"])
        return (self.flat_p[iw["off"]:iw["off"] + Ct * K].view(Ct, K), self.flat_g[iw["off"]:iw["off"] + Ct * K].view(Ct, K),
                self.flat_p[ib["off"]:ib["off"] + Ct], self.flat_g[ib["off"]:ib["off"] + Ct])

    # -- maintenance ---------------------------------------------------------------------------------------
    def attach_grads(self):
        for p in self.params:
            if not self.info[id(p)]["frozen"]:
                p.grad = self.grad_view(p)

    def refresh_shadow(self):
        ops.pack_shadow(self.flat_p, self.shadow, self.seg_dev, self.n_seg)
        self.shadow_dirty = False
        self.shadow_version += 1

    # -- transposed weight copies (bf16): dx = dy W reads W^T [K][N] as a plain K-contiguous operand ------------------
    def wt(self, w):
        """W^T for a 2-D view `w` of the bf16 shadow (single weights, fused q|k|v stacks, the packed K/V matrix), kept fresh by
        ONE batched transposition per shadow version; None when `w` is not such a view (f32 parity mode, odd shapes)."""
        if not DGRAD_PRETRANSPOSED or self.dtype != torch.bfloat16 or w.dim() != 2 or not w.is_contiguous():
            return None
        rows, cols = w.shape
        base = self.shadow.data_ptr()
        if rows % 64 or cols % 64 or not (base <= w.data_ptr() < base + 2 * self.n_shadow):
            return None
        key = (w.data_ptr(), rows, cols)
        e = self._wt.get(key)
        if e is None:
            dst = torch.empty(cols, rows, dtype=torch.bfloat16, device=self.device)
            e = self._wt[key] = (dst, [(w.data_ptr(), dst.data_ptr(), rows, cols, cols, rows)])
            return self._wt_register(e)
        self._wt_refresh()
        return e[0]

    def wd(self, w3, cout, cin_pad):
        """Data-gradient operand of a Conv1d k=3 weight shadow `w3` [Cout][3][cin_pad]: Wd [cin_pad][3*cout] with
        Wd[ci][t*cout + co] = W[co][2 - t][ci] (flipped taps, the reduction contiguous) -- three transposed sub-matrices."""
        if not DGRAD_PRETRANSPOSED or self.dtype != torch.bfloat16 or w3.dim() != 2 or not w3.is_contiguous():
            return None
        base = self.shadow.data_ptr()
        if cout % 64 or cin_pad % 64 or w3.shape[1] != 3 * cin_pad or cout > w3.shape[0] or \
                not (base <= w3.data_ptr() < base + 2 * self.n_shadow):
            return None
        key = (w3.data_ptr(), cout, -cin_pad)
        e = self._wt.get(key)
        if e is None:
            dst = torch.empty(cin_pad, 3 * cout, dtype=torch.bfloat16, device=self.device)
            segs = [(w3.data_ptr() + 2 * (2 - t) * cin_pad, dst.data_ptr() + 2 * t * cout, cout, cin_pad, 3 * cin_pad, 3 * cout)
                    for t in range(3)]
            e = self._wt[key] = (dst, segs)
            return self._wt_register(e)
        self._wt_refresh()
        return e[0]

    def _wt_register(self, e):
        self._wt_table = None
        if self._wt_version == self.shadow_version:              # the others are current: bring only the new one up to date
            ops.transpose_batch(*self._transpose_table([e]))
        else:
            self._wt_refresh()
        return e[0]

    def _wt_refresh(self):
        if self._wt_version != self.shadow_version:
            if self._wt_table is None:
                self._wt_table = self._transpose_table(list(self._wt.values()))
            ops.transpose_batch(*self._wt_table)
            self._wt_version = self.shadow_version

    def _transpose_table(self, entries):
        segs = [sg for _, sgs in entries for sg in sgs]
        arr = (L.pt_transpose_seg * len(segs))()
        tiles = 0
        for i, (src, dst, rows, cols, src_ld, dst_ld) in enumerate(segs):
            arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols = src, dst, rows, cols
            arr[i].src_ld, arr[i].dst_ld, arr[i].tile_begin = src_ld, dst_ld, tiles
            tiles += (rows // 64) * (cols // 64)
        return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(segs), tiles

    def enable_fp8(self, on=True):
        """BASELINE configs[4] ("fp8 MFMA GEMMs"): feed-forward GEMMs whose fp8 form is a net win (fp8_pays) read e4m3 weights /
        activations and e5m2 output gradients, per-tensor current scaling; everything else stays bf16."""
        if on and self.dtype != torch.bfloat16:
            raise ValueError("fp8 GEMMs need the bf16 activation path (dtype=torch.bfloat16)")
        self.fp8 = bool(on)

    def w8(self, p):
        """(W8 [N][K], W8^T [K][N], state) -- e4m3 copies of the bf16 shadow of a 2-D weight (kernel row order, i.e. GEGLU
        rows interleaved), re-quantised on first use after every optimizer step / shadow repack."""
        e = self._w8.get(id(p))
        if e is None:
            n, k = self.info[id(p)]["sshape"]
            e = dict(w8=torch.empty(n, k, dtype=torch.uint8, device=self.device),
                     w8t=torch.empty(k, n, dtype=torch.uint8, device=self.device),
                     state=torch.empty(L.PT_FP8_STATE_FLOATS, dtype=torch.float32, device=self.device), version=-1)
            self._w8[id(p)] = e
        if e["version"] != self.shadow_version:
            ops.fp8_quantize(self.w(p), e["w8"], e["state"], L.PT_FP8_E4M3, out_t=e["w8t"])
            e["version"] = self.shadow_version
        return e["w8"], e["w8t"], e["state"]

    def mark_dirty(self):
        """The master weights may have been written outside the fused AdamW kernel (load_state_dict, a torch optimizer
        stepping through the parameter views after loss.backward(), manual edits): the activation-dtype shadow the GEMM /
        conv kernels read must be repacked before the next forward.  An explicit flag, because `p.data = view` gives every
        Parameter its own version counter -- writes through the views never show up in flat_p._version."""
        self.shadow_dirty = True

    def ensure_shadow_fresh(self):
        if self.shadow_dirty:
            self.refresh_shadow()

    def zero_grad(self):
        self.flat_g.zero_()

    def adamw_step(self, lr, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6, max_norm=1.0, gnorm_sq=None):
        """Fused clip + AdamW + shadow refresh; `gnorm_sq` is a 1-element device tensor (computed here if None)."""
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.flat_p); self.adam_v = torch.zeros_like(self.flat_p)
        if gnorm_sq is None:
            gnorm_sq = torch.zeros(1, dtype=torch.float32, device=self.device)
            ops.sumsq(self.flat_g, gnorm_sq)
        self.step_count += 1
        self.ensure_shadow_fresh()       # frozen tensors keep their shadow: it must be current before the fused refresh
        ops.adamw_step(self.flat_p, self.flat_g, self.adam_m, self.adam_v, self.shadow, self.seg_dev, self.n_seg,
                       gnorm_sq, max_norm, lr, betas[0], betas[1], eps, weight_decay, self.step_count)
        self.shadow_dirty = False        # the kernel rewrote the shadow of every updated tensor
        self.shadow_version += 1
        return gnorm_sq


def adamw_step_sharded(st, reducer, lr, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6, max_norm=1.0):
    """The optimizer step of a data-parallel rank under parallel.ShardedGradReducer (after reducer.finish()): global gradient
    norm from the slice norms, clip + AdamW on the slices this rank owns (+ the redundantly updated bucket tails), the new values
    published through the gradient buffer, all-gathered, and adopted into master + shadow.  Returns the squared global norm."""
    if st.adam_m is None:
        st.adam_m = torch.zeros_like(st.flat_p); st.adam_v = torch.zeros_like(st.flat_p)
    own, tails, foreign = reducer.owned_ranges(), reducer.tail_ranges(), reducer.foreign_ranges()
    gnorm_sq = torch.zeros(1, dtype=torch.float32, device=st.device)
    for lo, hi in own:
        ops.sumsq(st.flat_g[lo:hi], gnorm_sq)
    if reducer.rank == 0:                        # the tails are identical everywhere: counted once
        for lo, hi in tails:
            ops.sumsq(st.flat_g[lo:hi], gnorm_sq)
    reducer.global_sum(gnorm_sq)
    st.step_count += 1
    st.ensure_shadow_fresh()
    for lo, hi in own:
        ops.adamw_step_range(st.flat_p, st.flat_g, st.adam_m, st.adam_v, st.shadow, st.seg_dev, st.n_seg, lo, hi, gnorm_sq, max_norm,
                             lr, betas[0], betas[1], eps, weight_decay, st.step_count, publish=True)
    for lo, hi in tails:
        if lo % 4:
            raise RuntimeError("sharded optimizer: bucket tails must start on a quad (ParamStore spans are 64-element aligned)")
        ops.adamw_step_range(st.flat_p, st.flat_g, st.adam_m, st.adam_v, st.shadow, st.seg_dev, st.n_seg, lo, hi, gnorm_sq, max_norm,
                             lr, betas[0], betas[1], eps, weight_decay, st.step_count, publish=False)
    reducer.allgather_published()
    for lo, hi in foreign:
        ops.import_params_range(st.flat_p, st.flat_g, st.shadow, st.seg_dev, st.n_seg, lo, hi)
    st.shadow_dirty = False
    st.shadow_version += 1
    return gnorm_sq


# ---------------------------------------------------------------------------------------------------------
# layer executors (x, dy are token-major 2-D tensors)
# ---------------------------------------------------------------------------------------------------------

# ---- weight-gradient GEMMs on a side stream --------------------------------------------------------------------------
# dgrad and wgrad of a layer both depend only on dy: the wgrad is issued on a second HIP stream so that it runs beside the
# dgrad chain (the GEMMs are operand-latency bound, two resident kernels keep more loads in flight).  Its result is needed
# only by the DP reducer / optimizer, which join the side stream first.
WGRAD_SIDE_STREAM = __import__("os").environ.get("PT_WGRAD_SIDE_STREAM", "1") != "0"
_side = {}
_DIAG_SKIP_WGRAD = __import__("os").environ.get("PT_DIAG_SKIP_WGRAD", "0") == "1"


N_SIDE = int(__import__("os").environ.get("PT_SIDE_STREAMS", "1"))     # single-launch wgrads rotate over this many side streams
_side_rr = [0]


def _side_stream0(device):
    st = _side.get((device, 0))
    if st is None:
        st = _side[(device, 0)] = torch.cuda.Stream(device=device)
    return st


def _side_stream(device):
    _side_rr[0] = (_side_rr[0] + 1) % N_SIDE
    key = (device, _side_rr[0])
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=device)
    return st


# The forward / dgrad chain is the critical path and the weight-gradient stream only fills what it leaves idle: the chain
# therefore runs on a HIGH-priority stream (this device offers priorities 0 and -1), so that its workgroups are dispatched
# ahead of queued wgrad workgroups instead of competing with them for every freed CU.
MAIN_HIGH_PRIORITY = __import__("os").environ.get("PT_MAIN_PRIORITY", "1") != "0"
_main = {}


def main_stream(device):
    """High-priority stream for the critical chain of a training step (None: stay on the caller's stream)."""
    if not MAIN_HIGH_PRIORITY:
        return None
    st = _main.get(device)
    if st is None:
        st = _main[device] = torch.cuda.Stream(device=device, priority=-1)
    return st


def on_side_stream(fn, *tensors, ordered=False):
    """Run fn() (kernel launches reading `tensors`) on the wgrad side stream, ordered after the current stream.
    ordered=True: always the SAME side stream (stream 0) -- the grouped weight-gradient launches share one slab workspace,
    their folds add into the gradients with plain read-modify-writes and deferred boundary corrections must land after their
    conv, so consecutive groups need strict stream order whatever PT_SIDE_STREAMS says."""
    if _DIAG_SKIP_WGRAD:                 # timing diagnostic only (wrong gradients): the main-stream chain by itself
        return
    if not WGRAD_SIDE_STREAM:
        fn()
        return
    dev = tensors[0].device
    side = _side_stream0(dev) if ordered else _side_stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
    for t in tensors:
        t.record_stream(side)           # keep the caching allocator from recycling them under the side kernels


def join_side_stream(device):
    flush_wgrads(device)
    cur = torch.cuda.current_stream(device)
    for (dev, _), st in _side.items():
        if dev == device:
            cur.wait_stream(st)


# ---- dropout keep masks ---------------------------------------------------------------------------------------------------
# The keep mask of a dropout site is drawn on the device with torch's generator (device RNG streams are never compared with
# the reference's); tests inject masks through `dropout_mask_hook(shape, p) -> uint8 tensor`.
dropout_mask_hook = [None]


def dropout_keep_mask(shape, p, device):
    if dropout_mask_hook[0] is not None:
        return dropout_mask_hook[0](tuple(shape), p).to(device=device, dtype=torch.uint8).contiguous()
    return (torch.rand(shape, device=device) >= p).to(torch.uint8)


# ---- cross-attention K/V reuse (inference) -----------------------------------------------------------------------------
# The K/V projections of a cross-attention layer depend only on the text-encoder output and the layer's weights.  Inside
# `with cross_kv_cache():` (forward-only callers whose conditioning and weights stay fixed: the reverse-diffusion sampler's
# 1000 denoiser calls, the autoregressive decode loop) each layer projects them once and reuses the buffer afterwards.
_kv_cache = [None]


class cross_kv_cache:
    def __enter__(self):
        self.prev = _kv_cache[0]
        _kv_cache[0] = {}
        return _kv_cache[0]

    def __exit__(self, *exc):
        _kv_cache[0] = self.prev
        return False


def cached_cross_kv(layer, ctx_in, compute):
    """compute() -> (k, v, kvbuf) for `layer` on `ctx_in`; memoised per layer while a cross_kv_cache is active."""
    cache = _kv_cache[0]
    if cache is None or torch.is_grad_enabled():
        return compute()
    key = id(layer)
    hit = cache.get(key)
    if hit is None or hit[0] is not ctx_in:
        hit = cache[key] = (ctx_in, compute())
    return hit[1]


# ---- batched cross-attention K/V (training) --------------------------------------------------------------------------------
# Unet1DConditionModel.fwd / bwd project K and V of ALL cross-attention layers with one GEMM over the packed weights
# (ParamStore.late_kv_views) and hand each layer its column slices here: {id(attn2 module): (k, v)} in forward,
# {id(attn2 module): (dk, dv)} -- slices of one d(kv) buffer -- in backward.
batched_kv = [None]
batched_dkv = [None]


# ---- grouped weight gradients ------------------------------------------------------------------------------------------
# bf16 weight gradients are not launched one by one: they are queued (descriptor + references to dy / x) and go out as ONE
# pt_wgrad_group launch per <= 8 problems (+ one fold launch) on the side stream -- the weight gradients of a transformer
# block or of a few resnets.  A weight gradient is a 4..32-tile output under a 8 192..32 768-token reduction: alone it needs
# 8-64 K-slices to fill the chip and ends in 64 MiB of split-K partials per launch; grouped, the same 256 workgroups carry
# 3-10 slices per problem and one set of partials per GROUP (csrc/gemm.hip: wgrad8p_group_kernel).  flush_wgrads() is called
# before a sub-module is announced to the data-parallel reducer and at the end of backward.
WGRAD_GROUPED = __import__("os").environ.get("PT_WGRAD_GROUPED", "1") != "0"
KV_BATCHED = __import__("os").environ.get("PT_KV_BATCHED", "1") != "0"
GEGLU_FUSED = WGRAD_GROUPED and __import__("os").environ.get("PT_GEGLU_FUSED", "1") != "0"   # needs the grouped wgrad's fold
CONV_WGRAD_FLAT = WGRAD_GROUPED and __import__("os").environ.get("PT_CONV_WGRAD_FLAT", "1") != "0"
WGRAD_GROUP_WGS = int(__import__("os").environ.get("PT_WGRAD_GROUP_WGS", "256"))


class _WgradQueue:
    def __init__(self, device):
        self.device = device
        self.lists = {0: [], 1: []}            # B operand class: 0 plain / concat, 1 conv gather
        self.deferred = {0: [], 1: []}         # problems that must NOT share a launch with the group being built (see add)
        self.ws = torch.empty(ops.wgrad_group_ws_floats(WGRAD_GROUP_WGS), dtype=torch.float32, device=device)

    def add(self, cls, desc, tiles, tensors, defer=False):
        """defer: the problem accumulates into a destination another problem of the CURRENT group also writes (the fold of one
        launch adds each problem's partials with plain read-modify-writes): it joins the next group of its class."""
        if defer:
            self.deferred[cls].append((desc, tiles, tensors))
            return
        lst = self.lists[cls]
        if lst and sum(t for _, t, _ in lst) + tiles > WGRAD_GROUP_WGS:
            self.flush(cls)
            lst = self.lists[cls]          # flush() starts a fresh list
        lst.append((desc, tiles, tensors))
        if len(lst) >= ops.WGRAD_GROUP_MAX:
            self.flush(cls)

    def flush(self, cls=None):
        """cls given: launch that class's current group (deferred problems then start the next one).  cls None (end of a
        backward pass): launch until nothing is queued."""
        for c in ((cls,) if cls is not None else (0, 1)):
            while True:
                lst = self.lists[c]
                if lst:
                    self.lists[c] = []
                    descs = [d for d, _, _ in lst]
                    tensors = [t for _, _, ts in lst for t in ts]
                    on_side_stream(lambda: ops.wgrad_group(descs, self.ws, WGRAD_GROUP_WGS), *tensors, ordered=True)
                pending, self.deferred[c] = self.deferred[c], []
                for item in pending:           # their conflicting partners have been launched: ordinary members from here on
                    self.add(c, *item)
                if cls is not None or not (self.lists[c] or self.deferred[c]):
                    break


_wq = {}


def _wgrad_queue(device):
    q = _wq.get(device)
    if q is None:
        q = _wq[device] = _WgradQueue(device)
    return q


def flush_wgrads(device=None):
    for dev, q in _wq.items():
        if device is None or dev == device:
            q.flush()


def _queue_wgrad(cls, M, N, K, A, B, gw, ldc, gbias, tensors, geglu_rows=0, alpha=1.0, defer=False):
    """Queue dW[M][N] += alpha A^T B for the grouped launch; False if this problem must take the single-launch path."""
    if not WGRAD_GROUPED or _DIAG_SKIP_WGRAD or tensors[0].dtype != torch.bfloat16 or M < 128 or N < 128:
        return False
    tiles = math.ceil(M / 256) * math.ceil(N / 256)
    if tiles > WGRAD_GROUP_WGS:
        return False
    kw = {}
    if gbias is not None:
        (gb,), n_rep, rstride = _rep(gbias)
        kw = dict(arow_sum=gb, arow_n=gbias.numel(), arow_rep=n_rep, arow_stride=rstride)
    desc = ops.gemm_desc(M, N, K, A, B, gw, ldc=ldc, out_kind=L.PT_OUT_F32_ATOMIC, geglu_rows=geglu_rows, alpha=alpha, **kw)
    _wgrad_queue(tensors[0].device).add(cls, desc, tiles, tuple(tensors), defer=defer)   # gw / gb live in the persistent flat buffers
    return True


def _empty(rows, cols, like):
    return torch.empty(rows, cols, dtype=like.dtype, device=like.device)


SPLITK_TARGET_WGS = int(__import__("os").environ.get("PT_SPLITK_WGS", "256"))


def _split_k(n_out, k_in, m_red, dtype):
    # every split adds one 128x128 f32 tile of atomics per output tile (~1.3 TB/s chip-wide): as few splits as still fill
    # the chip (the wgrad runs beside the dgrad chain on the side stream, so it need not fill all CUs by itself)
    tiles = math.ceil(n_out / 128) * math.ceil(k_in / 128)
    nkt = math.ceil(m_red / (64 if dtype == torch.bfloat16 else 32))
    return max(1, min(SPLITK_TARGET_WGS // max(tiles, 1), nkt // 4, 64))


def fp8_pays(M, N, K):
    """fp8 operands are a net win when the GEMM is large enough for the 256 x 256 fp8 kernel to run near 2x the bf16 rate AND
    the stand-alone quantisation of its M x K activation operand (two passes over M K bf16) is small beside it: wide outputs
    over a short reduction (measured at M = 8192, tools/fp8_probe.py: N = 8192, K = 1024 saves 48 us for 13 us of quantisation;
    N = 1024, K = 4096 saves 20 us for 45 us)."""
    return M >= 2048 and M % 256 == 0 and N >= 4 * K // 2 and N % 256 == 0 and K % 128 == 0


def fp8_quantize_act(x, fmt):
    """bf16 [M][K] -> (fp8 bytes, state) with current per-tensor scaling (state[1] = scale, device-resident)."""
    x8 = torch.empty(x.shape[0], x.shape[1], dtype=torch.uint8, device=x.device)
    state = torch.empty(L.PT_FP8_STATE_FLOATS, dtype=torch.float32, device=x.device)
    ops.fp8_quantize(x, x8, state, fmt)
    return x8, state


def linear_fwd(x, w, bias=None, residual=None, residual2=None, out=None, out_f32=False):
    M, K = x.shape
    N = w.shape[0]
    if (M <= 128 and N >= 2048 and K >= 512 and x.dtype == torch.float32 and out is None and residual is None
            and residual2 is None and _arena[0] is not None):      # fused training episode only: inference stays run-to-run bitwise
        # skinny f32 forward (the batched time-embedding projection: 32 rows against a [sum Cout][4 C0] weight, 130 MB): as one
        # 128-row tile per 128 columns it is 124 workgroups walking the whole reduction serially at 0.4 TB/s; split-K over
        # the reduction puts ~1000 workgroups on the weight stream (f32 atomics into the bias-initialised output)
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)      # never a view of the bias (expand(1, N) is contiguous)
        if bias is not None:
            out.copy_(bias.to(torch.float32).expand(M, N))
        else:
            out.zero_()
        ops.gemm(M, N, K, ops.plain(x), ops.plain(w), out, ops.pt_dtype(x), ldc=N, out_kind=L.PT_OUT_F32_ATOMIC,
                 split_k=max(1, min(8, K // 32 // 4)))
        return out
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    ops.gemm(M, N, K, ops.plain(x), ops.plain(w), out, ops.pt_dtype(x), ldc=out.stride(0),
             out_kind=L.PT_OUT_F32 if (out_f32 and x.dtype != torch.float32) else L.PT_OUT_T, bias=bias,
             residual=residual, ldr=residual.stride(0) if residual is not None else 0,
             residual2=residual2, ldr2=residual2.stride(0) if residual2 is not None else 0)
    return out


def linear_bwd(dy, x, w, gw, gbias=None, need_dx=True, dx_out=None, dx_accum=None, dx_residual=None, geglu_rows=0):
    """dy [M,N]; x [M,K]; w [N,K].  Returns dx (optionally dx = dy W + dx_residual, or accumulated in place).
    geglu_rows = F: dy's columns and w's rows are in the GEGLU-interleaved order; gw / gbias stay in the original order."""
    M, N = dy.shape
    K = x.shape[1]
    pt = ops.pt_dtype(x)
    def wgrad():
        kw = {}
        if gbias is not None:            # the bias gradient (column sums of dy) rides on the wgrad GEMM
            (gb,), n_rep, rstride = _rep(gbias)
            kw = dict(arow_sum=gb, arow_n=N, arow_rep=n_rep, arow_stride=rstride)
        ops.gemm(N, K, M, ops.plain(dy, trans=True), ops.plain(x, trans=True), gw, pt, ldc=gw.stride(0),
                 out_kind=L.PT_OUT_F32_ATOMIC, split_k=_split_k(N, K, M, x.dtype), **kw)
    if not _queue_wgrad(0, N, K, M, ops.plain(dy, trans=True), ops.plain(x, trans=True), gw, gw.stride(0), gbias, (dy, x),
                        geglu_rows=geglu_rows):
        if geglu_rows:
            raise RuntimeError("interleaved GEGLU weight gradients need the grouped wgrad path")
        on_side_stream(wgrad, dy, x)
    if not need_dx:
        return None
    if (M <= 128 and N >= 2048 and M % 4 == 0 and x.dtype == torch.float32 and dx_out is None and dx_accum is None
            and dx_residual is None):
        # skinny dgrad (the batched time-embedding projection: 32 rows, reduction over every resnet's channels): as
        # dx[M][K] it is 4 output tiles walking the whole reduction serially (0.5 ms); as dx^T = W^T dy^T it is a split-K
        # GEMM over the chip
        dy_t = dy.t().contiguous()
        dxt = torch.zeros(K, M, dtype=torch.float32, device=x.device)
        ops.gemm(K, M, N, ops.plain(w, trans=True), ops.plain(dy_t, trans=True), dxt, pt, ldc=M,
                 out_kind=L.PT_OUT_F32_ATOMIC, split_k=_split_k(K, M, N, x.dtype))
        return dxt.t().contiguous()
    dx = dx_out if dx_out is not None else (dx_accum if dx_accum is not None else _empty(M, K, x))
    res2 = dx_accum
    wt = transposed_weight(w)
    ops.gemm(M, K, N, ops.plain(dy), ops.plain(wt) if wt is not None else ops.plain(w, trans=True), dx, pt, ldc=dx.stride(0),
             residual=dx_residual, ldr=dx_residual.stride(0) if dx_residual is not None else 0,
             residual2=res2, ldr2=res2.stride(0) if res2 is not None else 0)
    return dx


_ROWMAP_DGRAD = {L.PT_MAP_S1: L.PT_MAP_S1, L.PT_MAP_S2: L.PT_MAP_S2_DGRAD}


def conv3_fwd(x, w3, bias, B, n_in, rowmap=L.PT_MAP_S1, cin=None, cout=None, row_bias=None, residual=None,
              x2=None, out=None, ldc=None, row_bias_ld=0):
    """Conv1d k=3 pad=1 as implicit GEMM.  x: (B*n_in, cin) token-major; w3: shadow [cout_pad][3*cin_pad]."""
    cin = x.shape[1] if cin is None else cin
    cout = w3.shape[0] if cout is None else cout
    n_out = {L.PT_MAP_S1: n_in, L.PT_MAP_S2: (n_in - 1) // 2 + 1, L.PT_MAP_UP2: 2 * n_in}[rowmap]
    M = B * n_out
    if out is None:
        out = _empty(M, cout, x)
    ops.gemm(M, cout, 3 * cin, ops.conv(x, cin, n_out, n_in, rowmap), ops.plain(w3), out, ops.pt_dtype(x),
             ldc=out.stride(0) if ldc is None else ldc, bias=bias, row_bias=row_bias, row_bias_rows=n_out, row_bias_ld=row_bias_ld,
             residual=residual, ldr=residual.stride(0) if residual is not None else 0)
    return out, n_out


def conv3_bwd(dy, x, w3, gw, gbias, B, n_in, n_out, rowmap=L.PT_MAP_S1, cin=None, cout=None, cin_store=None,
              need_dx=True, dx_residual=None):
    """dy: (B*n_out, cout[pad]); x: (B*n_in, cin).  gw: f32 [Cout][3*Cin_store] view of the flat grad buffer."""
    cin = x.shape[1] if cin is None else cin
    cout = dy.shape[1] if cout is None else cout           # channels the conv reads from dy (may be padded)
    pt = ops.pt_dtype(x)
    Mred = B * n_out
    padded = cin_store is not None and cin_store != cin

    def wgrad():
        kw = {}
        if gbias is not None:
            (gb,), n_rep, rstride = _rep(gbias)
            kw = dict(arow_sum=gb, arow_n=gbias.numel(), arow_rep=n_rep, arow_stride=rstride)
        ops.gemm(gw.shape[0], 3 * cin, Mred, ops.plain(dy, trans=True), ops.conv(x, cin, n_out, n_in, rowmap, trans=True),
                 gw, pt, ldc=3 * cin, out_kind=L.PT_OUT_F32_ATOMIC, split_k=_split_k(gw.shape[0], 3 * cin, Mred, x.dtype),
                 conv_wgrad_cin=cin if padded else 0, conv_wgrad_cin_store=cin_store if padded else 0, **kw)
    if CONV_WGRAD_FLAT and not padded and rowmap == L.PT_MAP_S1 and n_in == n_out and dy.dtype == torch.bfloat16 and \
            cin >= 128 and gw.shape[0] >= 128 and cin % 8 == 0 and dy.stride(0) == cout and dy.shape[1] == gw.shape[0]:
        # One FLAT item of B n rows: the gathered operand's addresses then advance by a constant step over the whole reduction
        # (per-item maps recompute them in the first two and the last k-tile of every item: 3 of 16 k-tiles at n = 1024 took
        # the slow path, 650 -> 830 TFLOP/s for the grouped launch).  The flat walk also multiplies each item's first output
        # row with the previous item's last input row (tap 0) and each item's last row with the next item's first (tap 2), where
        # the convolution pads with zeros: two rank-(B-1) products per conv, subtracted by two tiny problems of the plain group.
        ok = _queue_wgrad(1, gw.shape[0], 3 * cin, Mred, ops.plain(dy, trans=True),
                          ops.conv(x, cin, Mred, Mred, L.PT_MAP_S1, trans=True), gw, 3 * cin, gbias, (dy, x))
        if ok and B > 1:
            dy3, x3 = dy.view(B, n_out, dy.shape[1]), x.view(B, n_in, x.shape[1])
            first_dy, last_x = dy3[1:, 0, :], x3[:-1, n_in - 1, :cin]
            last_dy, first_x = dy3[:-1, n_out - 1, :], x3[1:, 0, :cin]
            # (the strided edge rows as one-tap "conv" operands, so that they ride in the SAME conv-class group as their conv)
            edge = lambda t: ops.conv(t, cin, B - 1, B - 1, L.PT_MAP_BACK, taps=1, trans=True)
            ok0 = _queue_wgrad(1, gw.shape[0], cin, B - 1, ops.plain(first_dy, trans=True), edge(last_x),
                               gw[:, :cin], 3 * cin, None, (dy, x), alpha=-1.0, defer=True)
            ok2 = _queue_wgrad(1, gw.shape[0], cin, B - 1, ops.plain(last_dy, trans=True), edge(first_x),
                               gw[:, 2 * cin:], 3 * cin, None, (dy, x), alpha=-1.0, defer=True)
            if not (ok0 and ok2):
                raise RuntimeError("conv weight gradient: the boundary corrections were refused by the grouped path")
        if not ok:
            on_side_stream(wgrad, dy, x)
    elif padded or not _queue_wgrad(1, gw.shape[0], 3 * cin, Mred, ops.plain(dy, trans=True),
                                    ops.conv(x, cin, n_out, n_in, rowmap, trans=True), gw, 3 * cin, gbias, (dy, x)):
        on_side_stream(wgrad, dy, x)
    if not need_dx:
        return None
    # the dgrad reads the flipped weights with the reduction (tap, cout) contiguous: a kept copy when the store has one
    wd = conv_dgrad_weight(w3, cout, cin) if w3.shape[1] == 3 * cin else None
    if rowmap == L.PT_MAP_UP2:
        # dgrad of (nearest x2 -> conv): stride-1 dgrad at the upsampled length, then fold row pairs
        dxu = _empty(B * 2 * n_in, cin, x)
        ops.gemm(B * 2 * n_in, cin, 3 * cout, ops.conv(dy, cout, 2 * n_in, n_out, L.PT_MAP_S1),
                 ops.plain(wd) if wd is not None else ops.wflip(w3, cout, cin), dxu, pt)
        dx = _empty(B * n_in, cin, x)
        ops.pairsum_rows(dxu, dx)
        return dx
    dx = _empty(B * n_in, cin, x)
    ops.gemm(B * n_in, cin, 3 * cout, ops.conv(dy, cout, n_in, n_out, _ROWMAP_DGRAD[rowmap]),
             ops.plain(wd) if wd is not None else ops.wflip(w3, cout, cin), dx, pt,
             residual=dx_residual, ldr=dx_residual.stride(0) if dx_residual is not None else 0)
    return dx


class GNState:
    __slots__ = ("mean", "rstd", "raw_eps")


N_REP = int(__import__("os").environ.get("PT_GRAD_REPLICAS", "16"))


class ZeroArena:
    """f32 scratch that is zero when handed out: ONE memset per training step (reset()) replaces a memset per GroupNorm
    statistics / workspace buffer (3 launches per GroupNorm forward+backward, ~180 per step).  It also holds the REPLICATED
    destinations of the small gradients that many workgroups accumulate with float atomics (biases, norm scales / shifts):
    N_REP zeroed copies each, summed into the flat gradient buffer by one pt_fold_replicas launch (fold())."""

    def __init__(self, device, n_floats=6 << 20):
        self.buf = torch.zeros(n_floats, dtype=torch.float32, device=device)
        self.off = 0
        self.dirty = False
        self.folds = []                  # (rep_off, rep_stride, dst data_ptr, n) not yet folded
        self._seg_cache = {}              # fold segment tables on the device, keyed by their contents (one per flush position)

    def reset(self):
        if self.folds:
            raise RuntimeError("ZeroArena.reset() with unfolded gradient replicas")
        if self.dirty:
            self.buf[:self.off].zero_()
        self.off = 0
        self.dirty = False

    def alloc(self, n):
        n_al = (n + 63) // 64 * 64
        if self.off + n_al > self.buf.numel():
            return torch.zeros(n, dtype=torch.float32, device=self.buf.device)
        out = self.buf[self.off:self.off + n]
        self.off += n_al
        self.dirty = True
        return out

    def replicated(self, *dsts):
        """Replica-0 views standing in for the 1-D f32 gradient views `dsts` (all the same length): (views, n_rep, stride)."""
        n = dsts[0].numel()
        n_al = (n + 63) // 64 * 64
        if N_REP <= 1 or self.off + len(dsts) * N_REP * n_al > self.buf.numel():
            return dsts, 1, 0
        out = []
        for d in dsts:
            self.folds.append((self.off, n_al, d.data_ptr(), n))
            out.append(self.buf[self.off:self.off + n])
            self.off += N_REP * n_al
        self.dirty = True
        return out, N_REP, n_al

    def fold(self, flat_g):
        """Sum every pending replica set into flat_g (the destinations are views of it)."""
        if not self.folds:
            return
        base = flat_g.data_ptr()
        key = tuple((ro, rs, (dp - base) // 4, n) for ro, rs, dp, n in self.folds)
        seg_dev = self._seg_cache.get(key)
        if seg_dev is None:
            raw = (L.pt_fold_seg * len(key))()
            for i, (ro, rs, do, n) in enumerate(key):
                if do < 0 or do + n > flat_g.numel():
                    raise RuntimeError("replicated destination is not a view of the flat gradient buffer")
                raw[i].rep_off, raw[i].rep_stride, raw[i].dst_off, raw[i].n = ro, rs, do, n
            if len(self._seg_cache) > 64:
                self._seg_cache.clear()
            seg_dev = self._seg_cache[key] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(flat_g.device)
        ops.fold_replicas(self.buf, flat_g, seg_dev, len(key), N_REP, max(k[3] for k in key))
        self.folds.clear()


_arena = [None]          # the arena of the fused forward+backward episode in progress (single-threaded host code)


def set_arena(a):
    _arena[0] = a


def _rep(*dsts):
    a = _arena[0]
    if a is None:
        return dsts, 1, 0
    return a.replicated(*dsts)


def fold_grad_replicas(flat_g):
    if _arena[0] is not None:
        _arena[0].fold(flat_g)


def groupnorm_fwd(x1, x2, gamma, beta, B, N, G, eps, silu, arena=None):
    C = x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
    s = GNState()
    y = _empty(B * N, C, x1)
    if x1.dtype == torch.bfloat16:   # one fused call (a single slab kernel for N <= 1024)
        s.mean = torch.empty(B * G, dtype=torch.float32, device=x1.device)
        s.rstd = torch.empty(B * G, dtype=torch.float32, device=x1.device)
        s.raw_eps = -1.0
        ops.groupnorm_fwd(x1, x2, gamma, beta, y, s.mean, s.rstd, B, N, G, eps, silu)
        return y, s
    if arena is not None:            # raw statistics in pre-zeroed scratch, finalized on the fly by every consumer
        sums = arena.alloc(2 * B * G)
        s.mean, s.rstd, s.raw_eps = sums[:B * G], sums[B * G:], float(eps)
        ops.groupnorm_stats(x1, x2, s.mean, s.rstd, B, N, G, -1.0)
    else:
        s.mean = torch.empty(B * G, dtype=torch.float32, device=x1.device)
        s.rstd = torch.empty(B * G, dtype=torch.float32, device=x1.device)
        s.raw_eps = -1.0
        ops.groupnorm_stats(x1, x2, s.mean, s.rstd, B, N, G, eps)
    ops.groupnorm_apply(x1, x2, s.mean, s.rstd, gamma, beta, y, None, B, N, G, silu, raw_eps=s.raw_eps)
    return y, s


def groupnorm_bwd(dy, x1, x2, s, gamma, beta, ggamma, gbeta, B, N, G, silu, dres=None, arena=None, item_sum=None):
    dx1 = torch.empty_like(x1)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    ws = arena.alloc(B * G * 2) if arena is not None else torch.empty(B * G * 2, dtype=torch.float32, device=x1.device)
    (ggamma, gbeta), n_rep, rstride = _rep(ggamma, gbeta)
    ops.groupnorm_bwd(dy, x1, x2, s.mean, s.rstd, gamma, beta, dres, dx1, dx2, ggamma, gbeta, ws, B, N, G, silu,
                      raw_eps=s.raw_eps, ws_zeroed=arena is not None, n_rep=n_rep, rep_stride=rstride, item_sum=item_sum)
    return dx1, dx2


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    M = x.shape[0]
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, eps)
    return y, (mean, rstd)


def layernorm_bwd(dy, x, stats, gamma, ggamma, gbeta, dres=None):
    dx = torch.empty_like(x)
    (ggamma, gbeta), n_rep, rstride = _rep(ggamma, gbeta)
    ops.layernorm_bwd(dy, x, stats[0], stats[1], gamma, dres, dx, ggamma, gbeta, n_rep=n_rep, rep_stride=rstride)
    return dx
