"""ResnetBlock1D / Upsample1D / Downsample1D on the HIP kernels (reference: tts/ldm/resnet.py:11-49,52-96,99-283).

Token-major throughout; the up-path's channel concat (unet_blocks.py:184,497) is never materialised: GroupNorm
reads the two sources directly and the 1x1 shortcut conv takes a concat operand.
"""
import torch
from torch import nn

from ... import _lib as L
from ... import engine as E
from ... import ops


class Upsample1D(nn.Module):
    """nearest x2 then Conv1d k3 (use_conv=True path, resnet.py:41-47), fused into one implicit GEMM."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.conv = nn.Conv1d(channels, channels, 3, padding=1)

    def fwd(self, st, x, B, N):
        y, n_out = E.conv3_fwd(x, st.w(self.conv.weight), st.f(self.conv.bias), B, N, L.PT_MAP_UP2, cout=self.channels)
        return y, (x, B, N, n_out)

    def bwd(self, st, saved, dy):
        x, B, N, n_out = saved
        return E.conv3_bwd(dy, x, st.w(self.conv.weight), st.g(self.conv.weight), st.g(self.conv.bias), B, N, n_out,
                           L.PT_MAP_UP2)


class Downsample1D(nn.Module):
    """Conv1d k3 stride 2 pad 1 (use_conv=True, padding=1 path, resnet.py:73,94)."""

    def __init__(self, channels, padding=1):
        super().__init__()
        if padding != 1:
            raise NotImplementedError("downsample_padding != 1 is not used by the reference's UNet")
        self.channels = channels
        self.conv = nn.Conv1d(channels, channels, 3, stride=2, padding=padding)

    def fwd(self, st, x, B, N):
        y, n_out = E.conv3_fwd(x, st.w(self.conv.weight), st.f(self.conv.bias), B, N, L.PT_MAP_S2, cout=self.channels)
        return y, (x, B, N, n_out)

    def bwd(self, st, saved, dy, dx_residual=None):
        x, B, N, n_out = saved
        return E.conv3_bwd(dy, x, st.w(self.conv.weight), st.g(self.conv.weight), st.g(self.conv.bias), B, N, n_out,
                           L.PT_MAP_S2, dx_residual=dx_residual)


class ResnetBlock1D(nn.Module):
    """GN->SiLU->conv3(+temb) -> GN->SiLU->conv3, + (1x1 conv | identity) shortcut, / output_scale_factor(=1)."""

    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6,
                 output_scale_factor=1.0, dropout=0.0, **_ignored):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if output_scale_factor != 1.0 or dropout != 0.0:
            raise NotImplementedError("output_scale_factor != 1 / dropout > 0 are not used by the reference's UNet")
        self.in_channels, self.out_channels, self.groups, self.eps = in_channels, out_channels, groups, eps
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv1d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv1d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def fwd(self, st, x1, x2, tproj, B, N):
        """x = concat(x1, x2) (x2 None on the down path); tproj = this block's (B, Cout) f32 column slice of the batched
        time-embedding projection (row stride = tproj.stride(0))."""
        Cin, Cout = self.in_channels, self.out_channels
        a1, s1 = E.groupnorm_fwd(x1, x2, st.f(self.norm1.weight), st.f(self.norm1.bias), B, N, self.groups, self.eps, True, arena=st.arena_active)
        h1, _ = E.conv3_fwd(a1, st.w(self.conv1.weight), st.f(self.conv1.bias), B, N, cin=Cin, cout=Cout, row_bias=tproj,
                            row_bias_ld=tproj.stride(0))
        a2, s2 = E.groupnorm_fwd(h1, None, st.f(self.norm2.weight), st.f(self.norm2.bias), B, N, self.groups, self.eps, True, arena=st.arena_active)
        if self.conv_shortcut is not None:
            M = B * N
            sc = torch.empty(M, Cout, dtype=x1.dtype, device=x1.device)
            A = ops.concat(x1, x2) if x2 is not None else ops.plain(x1)
            ops.gemm(M, Cout, Cin, A, ops.plain(st.w(self.conv_shortcut.weight)), sc, ops.pt_dtype(x1),
                     bias=st.f(self.conv_shortcut.bias))
            res = sc
        else:
            res = x1
        out, _ = E.conv3_fwd(a2, st.w(self.conv2.weight), st.f(self.conv2.bias), B, N, cin=Cout, cout=Cout, residual=res)
        return out, (x1, x2, a1, s1, h1, a2, s2, B, N)

    def bwd(self, st, saved, dout, dtproj):
        """Returns (dx1, dx2); writes d(time-embedding projection) into this block's (B, Cout) slice `dtproj` (zeroed)."""
        x1, x2, a1, s1, h1, a2, s2, B, N = saved
        Cin, Cout, M = self.in_channels, self.out_channels, B * N
        pt = ops.pt_dtype(x1)
        da2 = E.conv3_bwd(dout, a2, st.w(self.conv2.weight), st.g(self.conv2.weight), st.g(self.conv2.bias), B, N, N,
                          cin=Cout, cout=Cout)
        dh1, _ = E.groupnorm_bwd(da2, h1, None, s2, st.f(self.norm2.weight), st.f(self.norm2.bias),
                                 st.g(self.norm2.weight), st.g(self.norm2.bias), B, N, self.groups, True, arena=st.arena_active,
                                 item_sum=dtproj)
        # (item_sum: the time-embedding projection's gradient d tproj[b][c] = sum_n dh1[(b,n)][c] comes out of the GroupNorm
        # backward itself -- the slab kernel holds the whole item; its GEMMs are batched over all blocks by the UNet)
        # conv1.bias: its gradient is complete here, BEFORE the block is announced to the data-parallel reducer (it rides on
        # the wgrad GEMM's all-ones column like every other bias)
        da1 = E.conv3_bwd(dh1, a1, st.w(self.conv1.weight), st.g(self.conv1.weight), st.g(self.conv1.bias), B, N, N,
                          cin=Cin, cout=Cout)
        if self.conv_shortcut is not None:
            w, gw = st.w(self.conv_shortcut.weight), st.g(self.conv_shortcut.weight).view(Cout, Cin)
            A = ops.concat(x1, x2, trans=True) if x2 is not None else ops.plain(x1, trans=True)
            srcs = (dout, x1) + ((x2,) if x2 is not None else ())
            if not E._queue_wgrad(0, Cout, Cin, M, ops.plain(dout, trans=True), A, gw, Cin, st.g(self.conv_shortcut.bias), srcs):
                ops.gemm(Cout, Cin, M, ops.plain(dout, trans=True), A, gw, pt, out_kind=L.PT_OUT_F32_ATOMIC,
                         split_k=E._split_k(Cout, Cin, M, x1.dtype))
                ops.colsum(dout, st.g(self.conv_shortcut.bias), M, Cout)
            dres = torch.empty(M, Cin, dtype=x1.dtype, device=x1.device)
            wt = E.transposed_weight(w)                 # W^T [Cin][Cout] when the store keeps one (bf16)
            ops.gemm(M, Cin, Cout, ops.plain(dout), ops.plain(wt) if wt is not None else ops.plain(w, trans=True), dres, pt)
        else:
            dres = dout
        return E.groupnorm_bwd(da1, x1, x2, s1, st.f(self.norm1.weight), st.f(self.norm1.bias),
                               st.g(self.norm1.weight), st.g(self.norm1.bias), B, N, self.groups, True, dres=dres, arena=st.arena_active)
