"""Unet1DConditionModel on the HIP kernels (reference: tts/ldm/unet_1d_condition.py:111-412,553-739).

Same constructor surface for the arguments TTSSingleSpeaker forwards (tts/models.py:136-147); everything the
reference leaves at its default (act_fn silu, 32 groups, eps 1e-5, flip_sin_to_cos, freq_shift 0,
attention_head_dim 8 used as the NUMBER of heads) is fixed here the same way.
"""
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch
from torch import nn

from ... import _lib as L
from ... import engine as E
from ... import ops
from .unet_blocks import get_down_block, get_up_block, UNetMidBlock1DCrossAttn


@dataclass
class UNet1DConditionOutput:
    sample: torch.Tensor


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class Unet1DConditionModel(nn.Module):
    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str] = ("CrossAttnDownBlock1D", "DownBlock1D"),
                 mid_block_type: Optional[str] = "UNetMidBlock1DCrossAttn",
                 up_block_types: Tuple[str] = ("UpBlock1D", "CrossAttnUpBlock1D"),
                 block_out_channels: Tuple[int] = (320, 640), layers_per_block: Union[int, Tuple[int]] = 2,
                 norm_num_groups: int = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 1280,
                 attention_head_dim: int = 8):
        super().__init__()
        nb = len(down_block_types)
        if len(up_block_types) != nb:
            raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. "
                             f"`down_block_types`: {down_block_types}. `up_block_types`: {up_block_types}.")
        if len(block_out_channels) != nb:
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. "
                             f"`block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if not isinstance(layers_per_block, int) and len(layers_per_block) != nb:
            raise ValueError("Must provide the same number of `layers_per_block` as `down_block_types`.")
        boc = list(block_out_channels)
        lpb = [layers_per_block] * nb if isinstance(layers_per_block, int) else list(layers_per_block)
        self.cfg = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                        block_out_channels=boc, groups=norm_num_groups, eps=norm_eps,
                        flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift)
        self.cpad = E._round_up(max(in_channels, out_channels), 8)      # token-major I/O channel padding
        ted = boc[0] * 4
        self.conv_in = nn.Conv1d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        oc = boc[0]
        for i, t in enumerate(down_block_types):
            ic, oc = oc, boc[i]
            self.down_blocks.append(get_down_block(
                t, num_layers=lpb[i], in_channels=ic, out_channels=oc, temb_channels=ted, add_downsample=i != nb - 1,
                resnet_eps=norm_eps, resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                attn_num_head_channels=attention_head_dim))
        if mid_block_type == "UNetMidBlock1DCrossAttn":
            self.mid_block = UNetMidBlock1DCrossAttn(in_channels=boc[-1], temb_channels=ted, resnet_eps=norm_eps,
                                                     resnet_groups=norm_num_groups,
                                                     cross_attention_dim=cross_attention_dim,
                                                     attn_num_head_channels=attention_head_dim)
        elif mid_block_type is None:
            self.mid_block = None
        else:
            raise ValueError(f"unknown mid_block_type : {mid_block_type}")
        self.num_upsamplers = 0
        rboc, rlpb = boc[::-1], lpb[::-1]
        oc = rboc[0]
        for i, t in enumerate(up_block_types):
            prev, oc = oc, rboc[i]
            ic = rboc[min(i + 1, nb - 1)]
            last = i == nb - 1
            self.num_upsamplers += 0 if last else 1
            self.up_blocks.append(get_up_block(
                t, num_layers=rlpb[i] + 1, in_channels=ic, out_channels=oc, prev_output_channel=prev,
                temb_channels=ted, add_upsample=not last, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim, attn_num_head_channels=attention_head_dim))
        # forward-order list of every ResnetBlock1D and its column offset in the batched time-embedding projection;
        # this is also the order in which ParamStore packs the time_emb_proj tensors (module registration order)
        self._resnets = [m for m in self.modules() if m.__class__.__name__ == "ResnetBlock1D"]
        off = 0
        for r in self._resnets:
            r._tp_off = off
            off += r.out_channels
        self._tp_total = off
        self._cross_attn = [m.attn2 for m in self.modules() if m.__class__.__name__ == "BasicTransformerBlock" and m.attn2 is not None]
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv1d(boc[0], out_channels, 3, padding=1)

    # ---- forward over token-major tensors -----------------------------------------------------------------
    def fwd(self, st, xt, timesteps, ctx, B, T, S):
        """xt: (B*T, cpad) activation dtype; timesteps: int64 (B,); ctx: (B*S, d).  Returns pred (B*T, cpad)."""
        cfg = self.cfg
        factor = 2 ** self.num_upsamplers
        if T % factor != 0:
            raise ValueError(f"sample length {T} must be a multiple of {factor}: CrossAttnUpBlock1D ignores "
                             "upsample_size in the reference (unet_blocks.py:525-527), so other lengths cannot run")
        C0, n_q = cfg["block_out_channels"][0], cfg["in_channels"]
        dev = xt.device
        tape = {}
        # time embedding MLP in f32 on the master weights (tiny)
        t_emb = torch.empty(B, C0, dtype=torch.float32, device=dev)
        ops.timestep_embedding(timesteps, t_emb, cfg["flip_sin_to_cos"], float(cfg["freq_shift"]))
        te = self.time_embedding
        e1 = E.linear_fwd(t_emb, st.f(te.linear_1.weight), st.f(te.linear_1.bias))
        s1 = torch.empty_like(e1); ops.silu_fwd(e1, s1)
        emb = E.linear_fwd(s1, st.f(te.linear_2.weight), st.f(te.linear_2.bias))
        semb = torch.empty_like(emb); ops.silu_fwd(emb, semb)
        Wt, _, bt, _ = st.late_views()
        tproj_all = E.linear_fwd(semb, Wt, bt)                       # (B, sum Cout) f32: one GEMM for every resnet
        tp = lambda r: tproj_all[:, r._tp_off:r._tp_off + r.out_channels]
        tape["time"] = (t_emb, e1, s1, emb, semb)

        # K / V of every cross-attention layer in ONE GEMM over the packed weights (they all read `ctx`); each layer gets its
        # column slices through E.batched_kv.  Inside a cross_kv_cache (sampler: fixed conditioning) the product is reused.
        tape["kv"] = None
        kvp = st.late_kv_views()
        if kvp is not None:
            Wkv, _, where = kvp
            cache = E._kv_cache[0] if not torch.is_grad_enabled() else None
            hit = cache.get("kv_all") if cache is not None else None
            if hit is not None and hit[0] is ctx:
                kv_all = hit[1]
            else:
                kv_all = E.linear_fwd(ctx, Wkv)
                if cache is not None:
                    cache["kv_all"] = (ctx, kv_all)
            slices = {}
            for a in self._cross_attn:
                rk, rv, C = where[id(a.to_k.weight)], where[id(a.to_v.weight)], a.to_k.weight.shape[0]
                slices[id(a)] = (kv_all[:, rk:rk + C], kv_all[:, rv:rv + C])
            E.batched_kv[0] = slices
            tape["kv"] = (ctx, kv_all.shape[1])
        try:
            return self._fwd_blocks(st, xt, tp, tape, ctx, B, T, S, N0=T)
        finally:
            E.batched_kv[0] = None

    def _fwd_blocks(self, st, xt, tp, tape, ctx, B, T, S, N0):
        cfg = self.cfg
        C0 = cfg["block_out_channels"][0]
        dev = xt.device
        h, _ = E.conv3_fwd(xt, st.w(self.conv_in.weight), st.f(self.conv_in.bias), B, T, cin=self.cpad, cout=C0)
        tape["conv_in"] = xt
        skips = [(h, T)]
        N = T
        down = []
        for blk in self.down_blocks:
            rec = []
            for j, r in enumerate(blk.resnets):
                h, sv_r = r.fwd(st, h, None, tp(r), B, N)
                sv_a = None
                if blk.attentions is not None:
                    h, sv_a = blk.attentions[j].fwd(st, h, ctx, B, N, S)
                rec.append((sv_r, sv_a))
                skips.append((h, N))
            sv_d = None
            if blk.downsamplers is not None:
                h, sv_d = blk.downsamplers[0].fwd(st, h, B, N)
                N = sv_d[3]
                skips.append((h, N))
            down.append((rec, sv_d))
        tape["down"] = down
        if self.mid_block is not None:
            mb = self.mid_block
            h, m0 = mb.resnets[0].fwd(st, h, None, tp(mb.resnets[0]), B, N)
            h, m1 = mb.attentions[0].fwd(st, h, ctx, B, N, S)
            h, m2 = mb.resnets[1].fwd(st, h, None, tp(mb.resnets[1]), B, N)
            tape["mid"] = (m0, m1, m2)
        up = []
        for blk in self.up_blocks:
            rec = []
            for j, r in enumerate(blk.resnets):
                skip, n_skip = skips.pop()
                if n_skip != N:
                    raise RuntimeError("skip / hidden length mismatch")
                h, sv_r = r.fwd(st, h, skip, tp(r), B, N)
                sv_a = None
                if blk.attentions is not None:
                    h, sv_a = blk.attentions[j].fwd(st, h, ctx, B, N, S)
                rec.append((sv_r, sv_a))
            sv_u = None
            if blk.upsamplers is not None:
                h, sv_u = blk.upsamplers[0].fwd(st, h, B, N)
                N = sv_u[3]
            up.append((rec, sv_u))
        tape["up"] = up
        a, s = E.groupnorm_fwd(h, None, st.f(self.conv_norm_out.weight), st.f(self.conv_norm_out.bias), B, N,
                               cfg["groups"], cfg["eps"], True, arena=st.arena_active)
        pred = torch.zeros(B * T, self.cpad, dtype=xt.dtype, device=dev)
        E.conv3_fwd(a, st.w(self.conv_out.weight), st.f(self.conv_out.bias), B, T, cin=C0, cout=cfg["out_channels"],
                    out=pred, ldc=self.cpad)
        tape["out"] = (h, a, s)
        tape["dims"] = (B, T, S)
        return pred, tape

    # ---- backward -------------------------------------------------------------------------------------------
    def bwd(self, st, tape, dpred, on_ready=None):
        """dpred: (B*T, cpad) (pad channels zero).  Returns dctx (B*S, d).  `on_ready(module)` fires when every
        weight gradient of that sub-module is complete (the data-parallel reducer hangs its bucket all-reduce on it)."""
        cfg = self.cfg
        B, T, S = tape["dims"]
        C0 = cfg["block_out_channels"][0]
        t_emb, e1, s1, emb, semb = tape["time"]
        dtp_all = torch.zeros(B, self._tp_total, dtype=torch.float32, device=semb.device)
        dtp = lambda r: dtp_all[:, r._tp_off:r._tp_off + r.out_channels]
        notify = on_ready or (lambda m: None)

        dkv_all = None
        if tape.get("kv") is not None:
            ctx_kv, R = tape["kv"]
            _, _, where = st.late_kv_views()
            dkv_all = torch.empty(ctx_kv.shape[0], R, dtype=ctx_kv.dtype, device=ctx_kv.device)    # every slice is fully written
            E.batched_dkv[0] = {id(a): (dkv_all[:, where[id(a.to_k.weight)]:where[id(a.to_k.weight)] + a.to_k.weight.shape[0]],
                                        dkv_all[:, where[id(a.to_v.weight)]:where[id(a.to_v.weight)] + a.to_v.weight.shape[0]])
                                for a in self._cross_attn}
        try:
            dctx = self._bwd_blocks(st, tape, dpred, notify, dtp_all, dtp)
        finally:
            E.batched_dkv[0] = None
        if dkv_all is not None:
            # d(ctx) and the packed K/V weight gradient of all cross-attention layers: one dgrad, one (queued) wgrad
            Wkv, gWkv, _ = st.late_kv_views()
            dctx = E.linear_bwd(dkv_all, ctx_kv, Wkv, gWkv, dx_accum=dctx)
        return dctx

    def _bwd_blocks(self, st, tape, dpred, notify, dtp_all, dtp):
        cfg = self.cfg
        B, T, S = tape["dims"]
        C0 = cfg["block_out_channels"][0]
        t_emb, e1, s1, emb, semb = tape["time"]
        h, a, s = tape["out"]
        da = E.conv3_bwd(dpred, a, st.w(self.conv_out.weight), st.g(self.conv_out.weight), st.g(self.conv_out.bias),
                         B, T, T, cin=C0, cout=self.cpad)
        dh, _ = E.groupnorm_bwd(da, h, None, s, st.f(self.conv_norm_out.weight), st.f(self.conv_norm_out.bias),
                                st.g(self.conv_norm_out.weight), st.g(self.conv_norm_out.bias), B, T, cfg["groups"], True, arena=st.arena_active)
        notify(self.conv_out); notify(self.conv_norm_out)
        dctx = None
        dskips = []
        N = T
        for blk, (rec, sv_u) in zip(reversed(self.up_blocks), reversed(tape["up"])):
            if sv_u is not None:
                dh = blk.upsamplers[0].bwd(st, sv_u, dh)
                N = sv_u[2]
            for j in reversed(range(len(blk.resnets))):
                sv_r, sv_a = rec[j]
                if sv_a is not None:
                    dh, dctx = blk.attentions[j].bwd(st, sv_a, dh, dctx)
                dh, dskip = blk.resnets[j].bwd(st, sv_r, dh, dtp(blk.resnets[j]))
                dskips.append(dskip)
            notify(blk)
        if self.mid_block is not None:
            mb = self.mid_block
            m0, m1, m2 = tape["mid"]
            dh, _ = mb.resnets[1].bwd(st, m2, dh, dtp(mb.resnets[1]))
            dh, dctx = mb.attentions[0].bwd(st, m1, dh, dctx)
            dh, _ = mb.resnets[0].bwd(st, m0, dh, dtp(mb.resnets[0]))
            notify(mb)
        # down path: every block output was also a skip; dskips[i] is the up-path gradient of skips[i]
        for blk, (rec, sv_d) in zip(reversed(self.down_blocks), reversed(tape["down"])):
            if sv_d is not None:
                ds = dskips.pop()
                ops.add(dh, ds, dh)
                dh = blk.downsamplers[0].bwd(st, sv_d, dh)
            for j in reversed(range(len(blk.resnets))):
                sv_r, sv_a = rec[j]
                ds = dskips.pop()
                ops.add(dh, ds, dh)
                if sv_a is not None:
                    dh, dctx = blk.attentions[j].bwd(st, sv_a, dh, dctx)
                dh, _ = blk.resnets[j].bwd(st, sv_r, dh, dtp(blk.resnets[j]))
            notify(blk)
        ds = dskips.pop()
        ops.add(dh, ds, dh)
        xt = tape["conv_in"]
        E.conv3_bwd(dh, xt, st.w(self.conv_in.weight), st.g(self.conv_in.weight), st.g(self.conv_in.bias), B, T, T,
                    cin=self.cpad, cout=C0, cin_store=cfg["in_channels"], need_dx=False)
        notify(self.conv_in)
        # batched time-embedding projection backward: one dgrad + one wgrad + one column sum for all resnets.  The packed
        # time_emb_proj tensors live in the "late" region of the flat buffers, which no block span covers: the data-parallel
        # reducer picks that region up in finish(), after these writes.
        Wt, gWt, _, gbt = st.late_views()
        dsemb = E.linear_bwd(dtp_all, semb, Wt, gWt, None)
        for c0 in range(0, self._tp_total, 8192):                    # projection-bias gradients: column sums over the batch
            c1 = min(self._tp_total, c0 + 8192)
            ops.colsum(dtp_all[:, c0:c1], gbt[c0:c1], B, c1 - c0)
        # time-embedding MLP backward (f32)
        te = self.time_embedding
        demb = torch.empty_like(emb); ops.silu_bwd(dsemb, emb, demb)
        ds1 = E.linear_bwd(demb, s1, st.f(te.linear_2.weight), st.g(te.linear_2.weight), st.g(te.linear_2.bias))
        de1 = torch.empty_like(e1); ops.silu_bwd(ds1, e1, de1)
        E.linear_bwd(de1, t_emb, st.f(te.linear_1.weight), st.g(te.linear_1.weight), st.g(te.linear_1.bias), need_dx=False)
        notify(te)
        return dctx
