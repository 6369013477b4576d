"""The five UNet block types on the HIP kernels (reference: tts/ldm/unet_blocks.py:10-128,131-620).

Skip tensors are handed around as references; the up path never concatenates (see resnet.py).
`attention_mask` is accepted by the reference's blocks and never reaches attention (:359-398,603-617): there is
no mask parameter here at all.
"""
from torch import nn

from .resnet import ResnetBlock1D, Downsample1D, Upsample1D
from .transformer_1d import Transformer1DModel


def _attn(channels, heads, cross_attention_dim, groups):
    return Transformer1DModel(heads, channels // heads, in_channels=channels, num_layers=1,
                              cross_attention_dim=cross_attention_dim, norm_num_groups=groups)


class DownBlock1D(nn.Module):
    has_cross_attention = False

    def __init__(self, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps, resnet_groups, **_):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock1D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=temb_channels,
                                                    eps=resnet_eps, groups=resnet_groups) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample1D(out_channels)]) if add_downsample else None
        self.attentions = None


class CrossAttnDownBlock1D(nn.Module):
    has_cross_attention = True

    def __init__(self, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps, resnet_groups,
                 cross_attention_dim, attn_num_head_channels, **_):
        super().__init__()
        self.attentions = nn.ModuleList([_attn(out_channels, attn_num_head_channels, cross_attention_dim, resnet_groups)
                                         for _ in range(num_layers)])
        self.resnets = nn.ModuleList([ResnetBlock1D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=temb_channels,
                                                    eps=resnet_eps, groups=resnet_groups) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample1D(out_channels)]) if add_downsample else None


def _up_resnets(num_layers, in_channels, out_channels, prev_output_channel, temb_channels, eps, groups):
    rs = []
    for i in range(num_layers):
        skip = in_channels if i == num_layers - 1 else out_channels
        cin = prev_output_channel if i == 0 else out_channels
        rs.append(ResnetBlock1D(in_channels=cin + skip, out_channels=out_channels, temb_channels=temb_channels,
                                eps=eps, groups=groups))
    return nn.ModuleList(rs)


class UpBlock1D(nn.Module):
    has_cross_attention = False

    def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 resnet_eps, resnet_groups, **_):
        super().__init__()
        self.resnets = _up_resnets(num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                                   resnet_eps, resnet_groups)
        self.upsamplers = nn.ModuleList([Upsample1D(out_channels)]) if add_upsample else None
        self.attentions = None


class CrossAttnUpBlock1D(nn.Module):
    has_cross_attention = True

    def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 resnet_eps, resnet_groups, cross_attention_dim, attn_num_head_channels, **_):
        super().__init__()
        self.attentions = nn.ModuleList([_attn(out_channels, attn_num_head_channels, cross_attention_dim, resnet_groups)
                                         for _ in range(num_layers)])
        self.resnets = _up_resnets(num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                                   resnet_eps, resnet_groups)
        self.upsamplers = nn.ModuleList([Upsample1D(out_channels)]) if add_upsample else None


class UNetMidBlock1DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, resnet_eps, resnet_groups, cross_attention_dim,
                 attn_num_head_channels, **_):
        super().__init__()
        self.attentions = nn.ModuleList([_attn(in_channels, attn_num_head_channels, cross_attention_dim, resnet_groups)])
        self.resnets = nn.ModuleList([ResnetBlock1D(in_channels=in_channels, out_channels=in_channels,
                                                    temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups)
                                      for _ in range(2)])


_DOWN = {"CrossAttnDownBlock1D": CrossAttnDownBlock1D, "DownBlock1D": DownBlock1D}
_UP = {"CrossAttnUpBlock1D": CrossAttnUpBlock1D, "UpBlock1D": UpBlock1D}


def get_down_block(down_block_type, **kw):
    t = down_block_type[7:] if down_block_type.startswith("UNetRes") else down_block_type
    if t not in _DOWN:
        raise ValueError(f"{down_block_type} does not exist.")
    if t == "CrossAttnDownBlock1D" and kw.get("cross_attention_dim") is None:
        raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlock1D")
    return _DOWN[t](**kw)


def get_up_block(up_block_type, **kw):
    t = up_block_type[7:] if up_block_type.startswith("UNetRes") else up_block_type
    if t not in _UP:
        raise ValueError(f"{up_block_type} does not exist.")
    if t == "CrossAttnUpBlock1D" and kw.get("cross_attention_dim") is None:
        raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlock1D")
    return _UP[t](**kw)
