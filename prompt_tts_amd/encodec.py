"""Encodec 24 kHz code -> waveform decoder on the HIP kernels (reference: decode_codec.py:8-16).

Token-major activations; the transposed convs are plain GEMMs whose N = r*Cout output columns ARE the r new
time steps (zero-copy upsampling), the residual block's 1x1 conv and 1x1 shortcut are fused into one concat-K
GEMM, ELUs live in epilogues (or on fragment load), the 24 kHz end (<= 64 channels) runs on the row-streaming
kernel or the fused stage kernels, the LSTM recurrence as ONE persistent launch (bf16) or T+1 dependent launches (f32)
issued from inside the library; its status word is checked before a waveform is returned (LstmStatus).
Weights are a flat dict of EFFECTIVE tensors (weight-norm folded at load time; names in `WEIGHT_KEYS`).
"""
import ctypes as C

import torch

from . import _lib as L
from . import ops
from ._lib import lib, check

RATIOS = (8, 5, 4, 2)
WEIGHT_KEYS = ["codebooks", "conv0.w", "conv0.b"] + [f"lstm.{n}{l}" for l in range(2) for n in ("w_ih", "w_hh", "b_ih", "b_hh")] + \
    [f"{p}{i}.{s}" for i in range(4) for p, s in (("up", "w"), ("up", "b"))] + \
    [f"res{i}.{c}.{s}" for i in range(4) for c in ("c3", "c1", "sc") for s in ("w", "b")] + ["final.w", "final.b"]


def fold_weight_norm(g, v, dim=0):
    """w = g * v / ||v|| (norm over all dims but `dim`), the fold of torch.nn.utils.weight_norm."""
    dims = [d for d in range(v.dim()) if d != dim]
    return g * v / v.norm(2, dim=dims, keepdim=True)


def weights_from_encodec_state_dict(sd, n_q=8):
    """Map an `encodec.EncodecModel.state_dict()` (original package naming, weight_g/weight_v) to effective weights."""
    def conv(prefix, dim=0):
        return fold_weight_norm(sd[prefix + ".weight_g"], sd[prefix + ".weight_v"], dim), sd[prefix + ".bias"]
    W = {"codebooks": torch.stack([sd[f"quantizer.vq.layers.{q}._codebook.embed"] for q in range(n_q)])}
    W["conv0.w"], W["conv0.b"] = conv("decoder.model.0.conv.conv")
    for l in range(2):
        for n, k in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"), ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
            W[f"lstm.{n}{l}"] = sd[f"decoder.model.1.lstm.{k}_l{l}"]
    idx = 3
    for i in range(4):
        W[f"up{i}.w"], W[f"up{i}.b"] = conv(f"decoder.model.{idx}.convtr.convtr")
        rb = f"decoder.model.{idx + 1}"
        W[f"res{i}.c3.w"], W[f"res{i}.c3.b"] = conv(rb + ".block.1.conv.conv")
        W[f"res{i}.c1.w"], W[f"res{i}.c1.b"] = conv(rb + ".block.3.conv.conv")
        W[f"res{i}.sc.w"], W[f"res{i}.sc.b"] = conv(rb + ".shortcut.conv.conv")
        idx += 3
    W["final.w"], W["final.b"] = conv(f"decoder.model.{idx}.conv.conv")
    return W


def _conv_mat(w):
    """Conv1d (Cout,Cin,k) -> [Cout][k*Cin] with column = tap*Cin + ci."""
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def _pad_cols(m, mult):
    k = m.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return m.contiguous()
    out = torch.zeros(m.shape[0], kp, dtype=m.dtype)
    out[:, :k] = m
    return out


_SKIP_STATUS_CHECK = __import__("os").environ.get("PT_LSTM_CHECK", "1") == "0"


class LstmTimeout(RuntimeError):
    pass


def _retry_enabled():
    """PT_LSTM_RETRY=0 (read per call): raise at the first timed-out hand-off instead of retrying with the per-step kernels."""
    return __import__("os").environ.get("PT_LSTM_RETRY", "1") != "0"


class LstmStatus:
    """The status word of pt_lstm2_forward (include/prompt_tts_hip.h): cleared by the call, set by the persistent kernel when a
    hand-off between its workgroups timed out (not all of them resident, e.g. another stream's kernels holding CUs) -- the
    output is then garbage.  The word is copied to pinned host memory ON THE STREAM right behind the LSTM, so the rest of the
    decoder is enqueued without a stall; check() waits for that copy only and raises instead of returning a wrong waveform.
    One object serves ONE call at a time: `busy` is set when a call takes it and cleared by check(); a decoder hands a second,
    overlapping call (another host thread / stream) its own object instead of this one (_StatusPool)."""

    def __init__(self, device):
        self.dev_word = torch.zeros(1, dtype=torch.int32, device=device)
        self.host_word = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.busy = False

    def ptr(self):
        return self.dev_word.data_ptr()

    def fetch(self):
        self.host_word.copy_(self.dev_word, non_blocking=True)
        self.event.record(torch.cuda.current_stream(self.dev_word.device))

    def failed(self):
        """Waits for the copy of the word; True = a hand-off timed out.  Releases the object."""
        try:
            if _SKIP_STATUS_CHECK:       # PT_LSTM_CHECK=0: timing diagnostic only (tools/decode_probe.py)
                return False
            self.event.synchronize()
            return int(self.host_word[0]) != 0
        finally:
            self.busy = False

    def check(self):
        if self.failed():
            raise LstmTimeout("pt_lstm2_forward: a hand-off of the persistent LSTM timed out (its workgroups were not all "
                              "resident: is a kernel of another library holding CUs?) and so did the retry; the result was "
                              "discarded")


class _StatusPool:
    """Status words of one decoder / encoder: a call takes a free one (allocating a pinned word is slow, so they are kept) and
    gives it back in check().  Two overlapping calls on one object therefore never share a word -- round 3's silent failure:
    call B's clearing memset wiped the time-out call A's kernel had just raised, and A returned a garbage waveform."""

    def __init__(self, device):
        self.device, self.items, self.lock = device, [], __import__("threading").Lock()

    def take(self):
        with self.lock:
            for st in self.items:
                if not st.busy:
                    st.busy = True
                    return st
            st = LstmStatus(self.device)
            st.busy = True
            self.items.append(st)
            return st


def planes_from_f32(x):
    """(rows, C) f32 -> (rows, 2C) bf16 plane rows [hi | lo] (host-side weight preparation; small-input fallbacks)."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=1).contiguous()


def planes_to_f32(x):
    C2 = x.shape[1] // 2
    return x[:, :C2].float() + x[:, C2:].float()


def run_lstm2(B, T, x, xg0, w_hh0, wcat1, bias1, pt, device, dtype, status=None, exact_f32=False, per_step=False):
    """Two-layer LSTM + skip + ELU over (B*T, 512) token-major rows; returns (ELU(h1 + x), LstmStatus to check()).
    `status`: a reusable LstmStatus (pinned allocation is slow) that no other call is using.  per_step: the per-step kernels
    (the retry after a timed-out hand-off of the persistent form).  Every scratch buffer is allocated per call."""
    M = B * T
    h0 = torch.empty(M, 512, dtype=dtype, device=device); h1 = torch.empty_like(h0); ze = torch.empty_like(h0)
    c0 = torch.empty(B, 512, dtype=torch.float32, device=device); c1 = torch.empty_like(c0)
    st = status if status is not None else LstmStatus(device)
    ld = L.pt_lstm2_desc()
    ld.B, ld.T, ld.H = B, T, 512
    ld.x, ld.xg0, ld.whh0, ld.wcat1, ld.bias1 = x.data_ptr(), xg0.data_ptr(), w_hh0.data_ptr(), wcat1.data_ptr(), bias1.data_ptr()
    ld.h0_seq, ld.h1_seq, ld.c0, ld.c1, ld.out_elu = h0.data_ptr(), h1.data_ptr(), c0.data_ptr(), c1.data_ptr(), ze.data_ptr()
    ld.status = st.ptr()
    ld.exact_f32 = int(exact_f32)      # the encoder: its embeddings feed integer code decisions -> exact f32 MFMA
    ld.per_step = int(per_step)
    check(lib.pt_lstm2_forward(C.byref(ld), pt, ops._stream()), "pt_lstm2_forward")
    st.fetch()
    return ze, st


# f32 decode: products as a bf16 x 3 split on the bf16 MFMA (error ~2^-16 per product; the decode meets the 1e-3 bound with a
# wide margin) instead of the exact f32 MFMA, which runs at 1/16 of the bf16 rate.  PT_ENCODEC_F32_X3=0: exact f32 everywhere.
F32_X3 = __import__("os").environ.get("PT_ENCODEC_F32_X3", "1") != "0"
# ... and activations in SPLIT storage (PT_BF16X2: a row of C values = [C hi | C lo] bf16 planes, split once by the producer):
# a GEMM k-tile then carries both planes of 32 columns (bf16 x 3 products, no conversion in the loop), and the 3 kHz -> 24 kHz end runs as the f32-class
# fused stage kernels (csrc/encodec_x2.hip).  PT_ENCODEC_F32_PLANES=0: f32 activations, hi / lo split inside the product loops
# (round 3's path; kept for comparison).
F32_PLANES = __import__("os").environ.get("PT_ENCODEC_F32_PLANES", "1") != "0"
FUSED_TAIL = __import__("os").environ.get("PT_ENCODEC_FUSED_TAIL", "1") != "0"
FUSED_STAGES = __import__("os").environ.get("PT_ENCODEC_FUSED_STAGES", "1") != "0"


class EncodecDecoder:
    sample_rate = 24000

    def __init__(self, weights, device="cuda", dtype=torch.bfloat16):
        missing = [k for k in WEIGHT_KEYS if k not in weights]
        if missing:
            raise KeyError(f"missing decoder weights: {missing[:4]}...")
        self.device, self.dtype, self.pt = torch.device(device), dtype, ops._DT[dtype]
        self.x3 = F32_X3 and dtype == torch.float32
        self.x2 = self.x3 and F32_PLANES
        W = {k: v.detach().float().cpu() for k, v in weights.items()}
        d = lambda t: t.to(self.device, dtype).contiguous()
        f = lambda t: t.to(self.device, torch.float32).contiguous()
        self.n_q = W["codebooks"].shape[0]
        self.codebooks = d(W["codebooks"])
        self.w0, self.b0 = d(_conv_mat(W["conv0.w"])), f(W["conv0.b"])
        self.w_ih0 = d(W["lstm.w_ih0"]); self.bias0 = f(W["lstm.b_ih0"] + W["lstm.b_hh0"])
        self.w_hh0 = d(W["lstm.w_hh0"])
        self.wcat1 = d(torch.cat([W["lstm.w_ih1"], W["lstm.w_hh1"]], dim=1)); self.bias1 = f(W["lstm.b_ih1"] + W["lstm.b_hh1"])
        self.stages = []
        Cc = 512
        for i, r in enumerate(RATIOS):
            wt = W[f"up{i}.w"]                                            # (Cin, Cout, 2r)
            cin, cout = wt.shape[0], wt.shape[1]
            # Wt[rho*Cout + co][tap*Cin + ci] = w[ci][co][rho + tap*r]
            m = wt.view(cin, cout, 2, r).permute(3, 1, 2, 0).reshape(r * cout, 2 * cin)
            c3 = _conv_mat(W[f"res{i}.c3.w"])
            fused = torch.cat([W[f"res{i}.c1.w"][:, :, 0], W[f"res{i}.sc.w"][:, :, 0]], dim=1)
            small = cout <= 64                 # residual block on the row-streaming kernel
            small_up = r * cout <= 64          # transposed conv on the row-streaming kernel
            self.stages.append(dict(
                r=r, cin=cin, cout=cout, small=small, small_up=small_up,
                wt=d(_pad_cols(m, 32) if small_up else m), bt=f(W[f"up{i}.b"].repeat(r)),
                w3=d(_pad_cols(c3, 32) if small else c3), b3=f(W[f"res{i}.c3.b"]),
                wf=d(_pad_cols(fused, 32) if small else fused), bf=f(W[f"res{i}.c1.b"] + W[f"res{i}.sc.b"])))
            Cc = cout
        self.wfin, self.bfin = d(_pad_cols(_conv_mat(W["final.w"]), 32)), f(W["final.b"])
        if self.x2:
            # split storage: GEMM weights as plane rows [K hi | K lo] (the B operand of a PT_BF16X2 pt_gemm), the fused stage
            # kernels take the f32 matrices built above and split them per workgroup
            pl = lambda t: planes_from_f32(t.float().cpu()).to(self.device)
            self.w0_x2, self.w_ih0_x2 = pl(_conv_mat(W["conv0.w"])), pl(W["lstm.w_ih0"])
            for i, st in enumerate(self.stages):
                wt = W[f"up{i}.w"]
                cin, cout, r = st["cin"], st["cout"], st["r"]
                st["wt_x2"] = pl(wt.view(cin, cout, 2, r).permute(3, 1, 2, 0).reshape(r * cout, 2 * cin))
                st["w3_x2"] = pl(_conv_mat(W[f"res{i}.c3.w"]))
                st["wf_x2"] = pl(torch.cat([W[f"res{i}.c1.w"][:, :, 0], W[f"res{i}.sc.w"][:, :, 0]], dim=1))

    # -- helpers --------------------------------------------------------------------------------------------------
    def _lstm_status(self):
        pool = getattr(self, "_lstm_pool", None)
        if pool is None:
            pool = self._lstm_pool = _StatusPool(self.device)
        return pool.take()

    def _rowconv(self, Bn, n_rows, x, cin, taps, rowmap, w, bias, N, y, act=0, elu_x=0, x2=None, cin2=0, elu_x2=0, y_f32=False):
        d = L.pt_rowconv_desc()
        d.B, d.n_rows = Bn, n_rows
        d.x, d.ldx, d.cin, d.taps, d.rowmap, d.elu_x = x.data_ptr(), x.stride(0), cin, taps, rowmap, elu_x
        if x2 is not None:
            d.x2, d.ldx2, d.cin2, d.elu_x2 = x2.data_ptr(), x2.stride(0), cin2, elu_x2
        d.w, d.ldw, d.bias, d.N, d.act = w.data_ptr(), w.stride(0), bias.data_ptr(), N, act
        d.y, d.ldy, d.y_f32 = y.data_ptr(), y.stride(0), int(y_f32)
        d.f32_x3 = int(getattr(self, "x3", False))
        check(lib.pt_rowconv(C.byref(d), self.pt, ops._stream()), "pt_rowconv")

    def _empty(self, rows, cols, dtype=None):
        return torch.empty(rows, cols, dtype=dtype or self.dtype, device=self.device)

    # -- decode -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, codes):
        """codes (B, n_q, T) int64 in [0,1023] -> wav (B, 1, 320*T) f32.
        May be called from several host threads / streams at once: every scratch buffer and the status word belong to the call,
        and the library runs the persistent LSTM launches of one GPU one after the other (include/prompt_tts_hip.h) -- overlapping
        decodes therefore cost what sequential ones cost; more throughput = more GPUs.  A hand-off that times out anyway (kernels
        of another library holding CUs) is retried ONCE with the per-step kernels before anything is raised."""
        wav, lstm_status = self._decode(codes)
        if not _retry_enabled():
            lstm_status.check()
        elif lstm_status.failed():       # never hand out a waveform computed from a timed-out recurrence
            wav, lstm_status = self._decode(codes, per_step=True)
            lstm_status.check()
        return wav

    def _decode_x2(self, codes, per_step):
        """The f32-class decode on split storage (PT_BF16X2): every activation between launches is a bf16 plane row [C hi | C lo]."""
        B, n_q, T = codes.shape
        M, X2, bf = B * T, L.PT_BF16X2, torch.bfloat16
        e0 = self._empty(M, 256, bf)
        check(lib.pt_rvq_decode(codes.data_ptr(), self.codebooks.data_ptr(), e0.data_ptr(), B, n_q, T, 1024, 128, X2, ops._stream()), "pt_rvq_decode")
        y0 = self._empty(M, 1024, bf)
        ops.gemm(M, 512, 7 * 128, ops.conv(e0, 128, T, T, L.PT_MAP_CAUSAL_REFLECT, taps=7), ops.plain(self.w0_x2), y0, X2, ldc=1024, bias=self.b0)
        xg0 = self._empty(M, 2048, torch.float32)
        ops.gemm(M, 2048, 512, ops.plain(y0), ops.plain(self.w_ih0_x2), xg0, X2, out_kind=L.PT_OUT_F32, bias=self.bias0)
        # the recurrence: x / out as plane rows; where the persistent f32-class form does not apply (few rows, the per-step retry)
        # the planes are converted and the PT_F32 call is made instead
        st = self._lstm_status()
        ze = self._empty(M, 1024, bf)
        done = False
        if not per_step:
            h0 = torch.empty(M, 512, dtype=torch.float32, device=self.device)
            ld = L.pt_lstm2_desc()
            ld.B, ld.T, ld.H = B, T, 512
            ld.x, ld.xg0, ld.whh0, ld.wcat1, ld.bias1 = y0.data_ptr(), xg0.data_ptr(), self.w_hh0.data_ptr(), self.wcat1.data_ptr(), self.bias1.data_ptr()
            ld.h0_seq, ld.h1_seq, ld.c0, ld.c1, ld.out_elu = h0.data_ptr(), h0.data_ptr(), h0.data_ptr(), h0.data_ptr(), ze.data_ptr()
            ld.status = st.ptr()
            rc = lib.pt_lstm2_forward(C.byref(ld), X2, ops._stream())
            if rc == 0:
                st.fetch()
                done = True
            elif rc != -1:                       # PT_ERR_SHAPE = "this form does not apply here"; anything else is an error
                check(rc, "pt_lstm2_forward")
        if not done:
            zf, st = run_lstm2(B, T, planes_to_f32(y0), xg0, self.w_hh0, self.wcat1, self.bias1, L.PT_F32, self.device, torch.float32, st,
                               per_step=per_step)
            ze = planes_from_f32(zf)
        xe, n = ze, T
        for si, sg in enumerate(self.stages):
            r, cin, cout = sg["r"], sg["cin"], sg["cout"]
            Min, Mout, n_out = B * n, B * n * r, n * r
            if (r, cin, cout) == (2, 64, 32):
                wav = torch.empty(B * n_out, 1, dtype=torch.float32, device=self.device)
                td = L.pt_encodec_tail_desc()
                td.B, td.n, td.cin, td.cout, td.r = B, n, cin, cout, r
                td.x, td.ldx = xe.data_ptr(), xe.stride(0)
                td.wt, td.bt, td.w3, td.b3 = sg["wt"].data_ptr(), sg["bt"].data_ptr(), sg["w3"].data_ptr(), sg["b3"].data_ptr()
                td.wf, td.bf, td.wfin, td.bfin = sg["wf"].data_ptr(), sg["bf"].data_ptr(), self.wfin.data_ptr(), self.bfin.data_ptr()
                td.wav = wav.data_ptr()
                check(lib.pt_encodec_tail(C.byref(td), X2, ops._stream()), "pt_encodec_tail")
                return wav.view(B, 1, n_out), st
            if (r, cin, cout) == (4, 128, 64):
                oute = self._empty(Mout, 2 * cout, bf)
                sd = L.pt_encodec_stage_desc()
                sd.B, sd.n, sd.cin, sd.cout, sd.r = B, n, cin, cout, r
                sd.x, sd.ldx = xe.data_ptr(), xe.stride(0)
                sd.wt, sd.bt, sd.w3, sd.b3 = sg["wt"].data_ptr(), sg["bt"].data_ptr(), sg["w3"].data_ptr(), sg["b3"].data_ptr()
                sd.wf, sd.bf, sd.y, sd.ldy = sg["wf"].data_ptr(), sg["bf"].data_ptr(), oute.data_ptr(), oute.stride(0)
                check(lib.pt_encodec_stage(C.byref(sd), X2, ops._stream()), "pt_encodec_stage")
                xe, n = oute, n_out
                continue
            # transposed conv as a GEMM whose N = r cout columns are the r new rows: plane blocks of cout columns
            x1 = self._empty(Min, 2 * r * cout, bf)
            A = ops.conv(xe, cin, n, n, L.PT_MAP_BACK, taps=2)
            if cout == 128:                      # stage 1: raw x1 only, then the fused residual block
                ops.gemm(Min, r * cout, 2 * cin, A, ops.plain(sg["wt_x2"]), x1, X2, ldc=2 * r * cout, bias=sg["bt"], x2_block=cout)
                oute = self._empty(Mout, 2 * cout, bf)
                sd = L.pt_encodec_stage_desc()
                sd.B, sd.n, sd.cin, sd.cout, sd.r = B, n_out, cout, cout, 1
                sd.x, sd.ldx = x1.data_ptr(), 2 * cout
                sd.w3, sd.b3, sd.wf, sd.bf = sg["w3"].data_ptr(), sg["b3"].data_ptr(), sg["wf"].data_ptr(), sg["bf"].data_ptr()
                sd.y, sd.ldy = oute.data_ptr(), oute.stride(0)
                check(lib.pt_encodec_res(C.byref(sd), X2, ops._stream()), "pt_encodec_res")
                xe, n = oute, n_out
                continue
            # stage 0 (512 -> 256, r = 8): three GEMMs -- its residual block's weights (0.8 MiB as hi / lo) fit no CU
            x1e = self._empty(Min, 2 * r * cout, bf)
            ops.gemm(Min, r * cout, 2 * cin, A, ops.plain(sg["wt_x2"]), x1, X2, ldc=2 * r * cout, bias=sg["bt"], x2_block=cout,
                     out2=x1e, ldc2=2 * r * cout, act2=1)
            x1v, x1ev = x1.view(Mout, 2 * cout), x1e.view(Mout, 2 * cout)
            c3e = self._empty(Mout, cout, bf)            # cout / 2 channels as planes
            ops.gemm(Mout, cout // 2, 3 * cout, ops.conv(x1ev, cout, n_out, n_out, L.PT_MAP_CAUSAL_REFLECT, taps=3), ops.plain(sg["w3_x2"]),
                     c3e, X2, ldc=cout, bias=sg["b3"], act=1)
            oute = self._empty(Mout, 2 * cout, bf)
            cc = ops.concat(c3e, x1v); cc.c_split = cout // 2
            ops.gemm(Mout, cout, cout // 2 + cout, cc, ops.plain(sg["wf_x2"]), oute, X2, ldc=2 * cout, bias=sg["bf"], act=1)
            xe, n = oute, n_out
        raise RuntimeError("decoder stages do not end in the (2, 64, 32) tail")

    def _decode(self, codes, per_step=False):
        if codes.dim() != 3:
            raise BaseException("The encoded_frames must have the shape of [B, N_q, T]")
        if codes.shape[1] != self.n_q:
            raise ValueError(f"expected {self.n_q} codebooks, got {codes.shape[1]}")
        codes = codes.to(self.device, torch.int64).contiguous()
        B, n_q, T = codes.shape
        if T < 7:
            raise ValueError("the causal reflect padding of the k=7 convs needs at least 7 frames")
        if self.x2 and [(s["r"], s["cin"], s["cout"]) for s in self.stages] == [(8, 512, 256), (5, 256, 128), (4, 128, 64), (2, 64, 32)]:
            return self._decode_x2(codes, per_step)
        pt, M = self.pt, B * T
        e0 = self._empty(M, 128)
        check(lib.pt_rvq_decode(codes.data_ptr(), self.codebooks.data_ptr(), e0.data_ptr(), B, n_q, T, 1024, 128, pt,
                                ops._stream()), "pt_rvq_decode")
        y0 = self._empty(M, 512)
        ops.gemm(M, 512, 7 * 128, ops.conv(e0, 128, T, T, L.PT_MAP_CAUSAL_REFLECT, taps=7), ops.plain(self.w0), y0, pt, bias=self.b0, x3=self.x3)
        xg0 = self._empty(M, 2048)
        ops.gemm(M, 2048, 512, ops.plain(y0), ops.plain(self.w_ih0), xg0, pt, bias=self.bias0, x3=self.x3)
        ze, lstm_status = run_lstm2(B, T, y0, xg0, self.w_hh0, self.wcat1, self.bias1, pt, self.device, self.dtype, self._lstm_status(),
                                    per_step=per_step)
        xe, n = ze, T                     # xe = ELU(stage input), n = rows per batch item
        fuse_tail = (self.dtype == torch.bfloat16 and FUSED_TAIL and
                     (self.stages[-1]["r"], self.stages[-1]["cin"], self.stages[-1]["cout"]) == (2, 64, 32) and n * 160 >= 8)
        for si, st in enumerate(self.stages):
            r, cin, cout = st["r"], st["cin"], st["cout"]
            if fuse_tail and si == len(self.stages) - 1:
                # the 24 kHz end in ONE launch: transposed conv + residual block + final conv, intermediates in LDS
                wav = torch.empty(B * n * r, 1, dtype=torch.float32, device=self.device)
                td = L.pt_encodec_tail_desc()
                td.B, td.n, td.cin, td.cout, td.r = B, n, cin, cout, r
                td.x, td.ldx = xe.data_ptr(), xe.stride(0)
                td.wt, td.bt, td.w3, td.b3 = st["wt"].data_ptr(), st["bt"].data_ptr(), st["w3"].data_ptr(), st["b3"].data_ptr()
                td.wf, td.bf, td.wfin, td.bfin = st["wf"].data_ptr(), st["bf"].data_ptr(), self.wfin.data_ptr(), self.bfin.data_ptr()
                td.wav = wav.data_ptr()
                check(lib.pt_encodec_tail(C.byref(td), pt, ops._stream()), "pt_encodec_tail")
                return wav.view(B, 1, n * r), lstm_status
            if self.dtype == torch.bfloat16 and FUSED_STAGES and (r, cin, cout) == (4, 128, 64) and n >= 4:
                # transposed conv + residual block of the 3 kHz -> 12 kHz stage in one launch
                oute = self._empty(B * n * r, cout)
                sd = L.pt_encodec_stage_desc()
                sd.B, sd.n, sd.cin, sd.cout, sd.r = B, n, cin, cout, r
                sd.x, sd.ldx = xe.data_ptr(), xe.stride(0)
                sd.wt, sd.bt, sd.w3, sd.b3 = st["wt"].data_ptr(), st["bt"].data_ptr(), st["w3"].data_ptr(), st["b3"].data_ptr()
                sd.wf, sd.bf, sd.y, sd.ldy = st["wf"].data_ptr(), st["bf"].data_ptr(), oute.data_ptr(), oute.stride(0)
                check(lib.pt_encodec_stage(C.byref(sd), pt, ops._stream()), "pt_encodec_stage")
                xe, n = oute, n * r
                continue
            Min, Mout, n_out = B * n, B * n * r, n * r
            x1 = self._empty(Min, r * cout)
            x1e = None
            if self.dtype == torch.bfloat16 and FUSED_STAGES and cout == 128 and not st["small_up"] and n_out >= 3:
                # transposed conv as a GEMM (its 640 x 512 weights do not fit a CU), then the whole residual block in one launch
                ops.gemm(Min, r * cout, 2 * cin, ops.conv(xe, cin, n, n, L.PT_MAP_BACK, taps=2), ops.plain(st["wt"]), x1, pt, bias=st["bt"], x3=self.x3)
                oute = self._empty(Mout, cout)
                sd = L.pt_encodec_stage_desc()
                sd.B, sd.n, sd.cin, sd.cout, sd.r = B, n_out, cout, cout, 1
                sd.x, sd.ldx = x1.data_ptr(), cout
                sd.w3, sd.b3, sd.wf, sd.bf = st["w3"].data_ptr(), st["b3"].data_ptr(), st["wf"].data_ptr(), st["bf"].data_ptr()
                sd.y, sd.ldy = oute.data_ptr(), oute.stride(0)
                check(lib.pt_encodec_res(C.byref(sd), pt, ops._stream()), "pt_encodec_res")
                xe, n = oute, n_out
                continue
            if st["small_up"]:
                self._rowconv(B, n, xe, cin, 2, L.PT_MAP_BACK, st["wt"], st["bt"], r * cout, x1)
            else:
                x1e = self._empty(Min, r * cout)
                ops.gemm(Min, r * cout, 2 * cin, ops.conv(xe, cin, n, n, L.PT_MAP_BACK, taps=2), ops.plain(st["wt"]), x1, pt,
                         bias=st["bt"], out2=x1e, ldc2=r * cout, act2=1, x3=self.x3)
            x1v = x1.view(Mout, cout)
            c3e = self._empty(Mout, cout // 2)
            oute = self._empty(Mout, cout)
            if st["small"]:
                src, elu = (x1e.view(Mout, cout), 0) if x1e is not None else (x1v, 1)
                self._rowconv(B, n_out, src, cout, 3, L.PT_MAP_CAUSAL_REFLECT, st["w3"], st["b3"], cout // 2, c3e, act=1, elu_x=elu)
                self._rowconv(B, n_out, c3e, cout // 2, 1, L.PT_MAP_BACK, st["wf"], st["bf"], cout, oute, act=1,
                              x2=x1v, cin2=cout)
            else:
                ops.gemm(Mout, cout // 2, 3 * cout, ops.conv(x1e.view(Mout, cout), cout, n_out, n_out, L.PT_MAP_CAUSAL_REFLECT, taps=3),
                         ops.plain(st["w3"]), c3e, pt, bias=st["b3"], act=1, x3=self.x3)
                ops.gemm(Mout, cout, cout // 2 + cout, ops.concat(c3e, x1v), ops.plain(st["wf"]), oute, pt, bias=st["bf"], act=1, x3=self.x3)
            xe, n = oute, n_out
        wav = torch.empty(B * n, 1, dtype=torch.float32, device=self.device)
        self._rowconv(B, n, xe, 32, 7, L.PT_MAP_CAUSAL_REFLECT, self.wfin, self.bfin, 1, wav, y_f32=True)
        return wav.view(B, 1, n), lstm_status


# =====================================================================================================================
# ENCODE: waveform -> codes (reference: data_preparation/generate_code.py:45-51, encodec `model.encode`; SURVEY a-12)
# =====================================================================================================================
ENC_RATIOS = (2, 4, 5, 8)
ENC_WEIGHT_KEYS = ["codebooks", "enc.conv0.w", "enc.conv0.b"] + \
    [f"enc.res{i}.{c}.{s}" for i in range(4) for c in ("c3", "c1", "sc") for s in ("w", "b")] + \
    [f"enc.down{i}.{s}" for i in range(4) for s in ("w", "b")] + \
    [f"enc.lstm.{n}{l}" for l in range(2) for n in ("w_ih", "w_hh", "b_ih", "b_hh")] + ["enc.final.w", "enc.final.b"]


def encoder_weights_from_encodec_state_dict(sd, n_q=8):
    """Map an `encodec.EncodecModel.state_dict()` (original package naming) to effective ENCODER weights + codebooks."""
    def conv(prefix):
        return fold_weight_norm(sd[prefix + ".weight_g"], sd[prefix + ".weight_v"], 0), sd[prefix + ".bias"]
    W = {"codebooks": torch.stack([sd[f"quantizer.vq.layers.{q}._codebook.embed"] for q in range(n_q)])}
    W["enc.conv0.w"], W["enc.conv0.b"] = conv("encoder.model.0.conv.conv")
    idx = 1
    for i in range(4):
        rb = f"encoder.model.{idx}"
        W[f"enc.res{i}.c3.w"], W[f"enc.res{i}.c3.b"] = conv(rb + ".block.1.conv.conv")
        W[f"enc.res{i}.c1.w"], W[f"enc.res{i}.c1.b"] = conv(rb + ".block.3.conv.conv")
        W[f"enc.res{i}.sc.w"], W[f"enc.res{i}.sc.b"] = conv(rb + ".shortcut.conv.conv")
        W[f"enc.down{i}.w"], W[f"enc.down{i}.b"] = conv(f"encoder.model.{idx + 2}.conv.conv")
        idx += 3
    for l in range(2):
        for n, k in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"), ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
            W[f"enc.lstm.{n}{l}"] = sd[f"encoder.model.{idx}.lstm.{k}_l{l}"]
    W["enc.final.w"], W["enc.final.b"] = conv(f"encoder.model.{idx + 2}.conv.conv")
    return W


class EncodecEncoder:
    """SEANet encoder + residual vector quantiser on the HIP kernels, token-major.  conv k7 1->32 and the 32/64-channel
    layers (the 24 kHz end, HBM-bound) run on the row-streaming kernel, everything from 128 channels on pt_gemm with the
    strided causal-reflect row map (implicit GEMM: no im2col, no padded copies); the LSTM is the decoder's; each RVQ stage
    is an exact-f32 GEMM (scores = 2 x.e - |e|^2) plus a one-wave-per-token argmax / residual-update kernel.
    The default dtype is float32: code indices are decided by gaps far below bf16 resolution."""
    sample_rate = 24000
    hop = 320

    def __init__(self, weights, device="cuda", dtype=torch.float32):
        missing = [k for k in ENC_WEIGHT_KEYS if k not in weights]
        if missing:
            raise KeyError(f"missing encoder weights: {missing[:4]}...")
        self.device, self.dtype, self.pt = torch.device(device), dtype, ops._DT[dtype]
        W = {k: v.detach().float().cpu() for k, v in weights.items()}
        d = lambda t: t.to(self.device, dtype).contiguous()
        f = lambda t: t.to(self.device, torch.float32).contiguous()
        self.n_q = W["codebooks"].shape[0]
        self.cb = f(W["codebooks"])                                     # (n_q, 1024, 128) f32
        self.cb_bias = f(-(W["codebooks"] ** 2).sum(-1))                # -|e|^2
        w0 = torch.zeros(32, 7, 8); w0[:, :, 0] = W["enc.conv0.w"][:, 0, :]          # the sample sits in channel 0 of 8
        self.w0, self.b0 = d(_pad_cols(w0.reshape(32, 56), 32)), f(W["enc.conv0.b"])
        self.stages = []
        Cc = 32
        for i, r in enumerate(ENC_RATIOS):
            c3 = _conv_mat(W[f"enc.res{i}.c3.w"])
            fused = torch.cat([W[f"enc.res{i}.c1.w"][:, :, 0], W[f"enc.res{i}.sc.w"][:, :, 0]], dim=1)
            dn = _conv_mat(W[f"enc.down{i}.w"])                          # [2C][2r*C], column = tap*C + ci
            small = Cc <= 64                      # residual block on the row-streaming kernel
            small_dn = 2 * Cc <= 64               # strided conv on the row-streaming kernel
            self.stages.append(dict(
                r=r, C=Cc, small=small, small_dn=small_dn,
                w3=d(_pad_cols(c3, 32) if small else c3), b3=f(W[f"enc.res{i}.c3.b"]),
                wf=d(_pad_cols(fused, 32) if small else fused), bf=f(W[f"enc.res{i}.c1.b"] + W[f"enc.res{i}.sc.b"]),
                wd=d(_pad_cols(dn, 32) if small_dn else dn), bd=f(W[f"enc.down{i}.b"])))
            Cc *= 2
        self.w_ih0 = d(W["enc.lstm.w_ih0"]); self.bias0 = f(W["enc.lstm.b_ih0"] + W["enc.lstm.b_hh0"])
        self.w_hh0 = d(W["enc.lstm.w_hh0"])
        self.wcat1 = d(torch.cat([W["enc.lstm.w_ih1"], W["enc.lstm.w_hh1"]], dim=1))
        self.bias1 = f(W["enc.lstm.b_ih1"] + W["enc.lstm.b_hh1"])
        self.wfin, self.bfin = d(_conv_mat(W["enc.final.w"])), f(W["enc.final.b"])

    _rowconv = EncodecDecoder._rowconv
    _lstm_status = EncodecDecoder._lstm_status
    _empty = EncodecDecoder._empty

    def _rowconv_strided(self, Bn, n_out, x, cin, taps, stride, w, bias, N, y):
        d = L.pt_rowconv_desc()
        d.B, d.n_rows = Bn, n_out
        d.x, d.ldx, d.cin, d.taps, d.rowmap, d.elu_x, d.stride = x.data_ptr(), x.stride(0), cin, taps, L.PT_MAP_STRIDED_REFLECT, 0, stride
        d.w, d.ldw, d.bias, d.N, d.act = w.data_ptr(), w.stride(0), bias.data_ptr(), N, 0
        d.y, d.ldy, d.y_f32 = y.data_ptr(), y.stride(0), 0
        check(lib.pt_rowconv(C.byref(d), self.pt, ops._stream()), "pt_rowconv")

    @torch.no_grad()
    def embeddings(self, wav):
        """wav (B, 1, L) f32, L a multiple of 320 -> (B*T, 128) f32 token-major embeddings, T = L / 320.
        A timed-out hand-off of the persistent LSTM is retried once with the per-step kernels (see EncodecDecoder.decode)."""
        emb, B, T, st = self._embeddings(wav)
        if not _retry_enabled():
            st.check()
        elif st.failed():
            emb, B, T, st = self._embeddings(wav, per_step=True)
            st.check()
        return emb, B, T

    def _embeddings(self, wav, per_step=False):
        if wav.dim() != 3 or wav.shape[1] != 1:
            raise ValueError("wav must be (B, 1, L)")
        B, _, Ln = wav.shape
        if Ln % self.hop != 0 or Ln < 7 * self.hop:
            raise ValueError("the length must be a multiple of 320 samples and at least 7 frames")
        pt = self.pt
        x0 = torch.zeros(B * Ln, 8, dtype=self.dtype, device=self.device)
        x0[:, 0] = wav.to(self.device, self.dtype).reshape(-1)
        n = Ln
        cur = self._empty(B * n, 32)
        self._rowconv(B, n, x0, 8, 7, L.PT_MAP_CAUSAL_REFLECT, self.w0, self.b0, 32, cur)
        cur_e = None                         # ELU(cur) when the producer could write it as a second output
        for i, st in enumerate(self.stages):
            r, Cc = st["r"], st["C"]
            M = B * n
            c3e = self._empty(M, Cc // 2); oute = self._empty(M, Cc)
            if st["small"]:
                self._rowconv(B, n, cur, Cc, 3, L.PT_MAP_CAUSAL_REFLECT, st["w3"], st["b3"], Cc // 2, c3e, act=1, elu_x=1)
                self._rowconv(B, n, c3e, Cc // 2, 1, L.PT_MAP_BACK, st["wf"], st["bf"], Cc, oute, act=1, x2=cur, cin2=Cc)
            else:
                ops.gemm(M, Cc // 2, 3 * Cc, ops.conv(cur_e, Cc, n, n, L.PT_MAP_CAUSAL_REFLECT, taps=3), ops.plain(st["w3"]), c3e, pt,
                         bias=st["b3"], act=1)
                ops.gemm(M, Cc, Cc // 2 + Cc, ops.concat(c3e, cur), ops.plain(st["wf"]), oute, pt, bias=st["bf"], act=1)
            n_out = n // r
            nxt = self._empty(B * n_out, 2 * Cc)
            nxt_e = None
            if st["small_dn"]:
                self._rowconv_strided(B, n_out, oute, Cc, 2 * r, r, st["wd"], st["bd"], 2 * Cc, nxt)
            else:
                need_e = i + 1 < len(self.stages) and not self.stages[i + 1]["small"]   # next residual block runs on pt_gemm: it reads ELU(x)
                if need_e:
                    nxt_e = self._empty(B * n_out, 2 * Cc)
                ops.gemm(B * n_out, 2 * Cc, 2 * r * Cc, ops.conv(oute, Cc, n_out, n, L.PT_MAP_STRIDED_REFLECT, taps=2 * r, stride=r),
                         ops.plain(st["wd"]), nxt, pt, bias=st["bd"], out2=nxt_e, ldc2=2 * Cc, act2=1)
            cur, cur_e, n = nxt, nxt_e, n_out
        T = n
        M = B * T
        xg0 = self._empty(M, 2048)
        ops.gemm(M, 2048, 512, ops.plain(cur), ops.plain(self.w_ih0), xg0, pt, bias=self.bias0)
        ze, lstm_status = run_lstm2(B, T, cur, xg0, self.w_hh0, self.wcat1, self.bias1, pt, self.device, self.dtype, self._lstm_status(),
                                    exact_f32=True, per_step=per_step)
        emb = torch.empty(M, 128, dtype=torch.float32, device=self.device)
        ops.gemm(M, 128, 7 * 512, ops.conv(ze, 512, T, T, L.PT_MAP_CAUSAL_REFLECT, taps=7), ops.plain(self.wfin), emb, pt,
                 bias=self.bfin, out_kind=L.PT_OUT_F32 if self.dtype != torch.float32 else L.PT_OUT_T)
        return emb, B, T, lstm_status

    @torch.no_grad()
    def quantize(self, emb, B, T):
        """emb (B*T, 128) f32 -> codes (B, n_q, T) int64: n_q stages of exact-f32 scores GEMM + argmax / residual update."""
        M = B * T
        res = emb.clone()
        codes = torch.empty(B, self.n_q, T, dtype=torch.int64, device=self.device)
        scores = torch.empty(M, 1024, dtype=torch.float32, device=self.device)
        for q in range(self.n_q):
            ops.gemm(M, 1024, 128, ops.plain(res), ops.plain(self.cb[q]), scores, L.PT_F32, bias=self.cb_bias[q], alpha=2.0)
            check(lib.pt_rvq_search(scores.data_ptr(), self.cb[q].data_ptr(), res.data_ptr(), codes.data_ptr(), B, self.n_q, T, q,
                                    1024, 128, ops._stream()), "pt_rvq_search")
        return codes

    @torch.no_grad()
    def encode(self, wav):
        """wav (B, 1, L) f32 -> codes (B, n_q, L/320) int64 (generate_code.py:48)."""
        emb, B, T = self.embeddings(wav)
        return self.quantize(emb, B, T)
