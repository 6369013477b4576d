"""a GEGLU projection weight was marked row-interleaved whenever F2 % 512 == 0 and d % 8 == 0, but its weight
gradient can only be un-interleaved by the GROUPED wgrad launch, which refuses d < 128 and more than WGRAD_GROUP_WGS tiles -- bf16
training then crashed for d = 64 and d >= 1536.  Eligibility now IS the grouped launch's predicate."""
import math

import pytest
import torch


def test_interleave_predicate_matches_the_grouped_wgrad_limits():
    from prompt_tts_amd import engine as E
    assert E.geglu_interleave_ok(8 * 512, 512)            # configs[1]
    assert E.geglu_interleave_ok(8 * 1024, 1024)          # configs[4]: 32 x 4 tiles
    assert E.geglu_interleave_ok(8 * 256, 256)            # configs[0]
    assert not E.geglu_interleave_ok(512, 64)             # d < 128: the grouped launch refuses N < 128
    assert not E.geglu_interleave_ok(8 * 1536, 1536)      # 48 x 6 = 288 tiles > 256 workgroups
    assert not E.geglu_interleave_ok(8 * 72, 72)          # F2 not a multiple of 512
    for d in (64, 128, 192, 256, 512, 768, 1024, 1280, 1536, 2048):
        tiles = math.ceil(8 * d / 256) * math.ceil(d / 256)
        assert E.geglu_interleave_ok(8 * d, d) == (d >= 128 and (8 * d) % 512 == 0 and tiles <= E.WGRAD_GROUP_WGS)


@pytest.mark.gpu
def test_small_text_width_trains_in_bf16(dev):
    """cross_attention_dim = 64: the text encoder's GEGLU projection is [512][64] -- one bf16 training step must run (plain layout,
    stand-alone GEGLU) and agree with the f32 parity mode."""
    from prompt_tts_amd.tts.models import TTSSingleSpeaker
    cfg = {"cmu_vocab_len": 149, "cmu_seq_len": 32, "cross_attention_dim": 64, "attention_head_dim": 64,
           "text_encoder_dropout": 0.0, "text_encoder_layers": 2, "sample_size": 64, "in_channels": 2, "out_channels": 2,
           "layers_per_block": 1, "block_out_channels": [256, 256], "down_block_types": ["CrossAttnDownBlock1D", "DownBlock1D"],
           "mid_block_type": "UNetMidBlock1DCrossAttn", "up_block_types": ["UpBlock1D", "CrossAttnUpBlock1D"]}
    g = torch.Generator().manual_seed(5)
    B = 4
    batch = [torch.rand(B, 2, 64, generator=g) * 2 - 1, torch.randn(B, 2, 64, generator=g), torch.randint(0, 1000, (B,), generator=g),
             torch.randint(1, 149, (B, 32), generator=g, dtype=torch.int32), torch.ones(B, 32, dtype=torch.int32)]
    batch = [x.to(dev) for x in batch]
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        m = TTSSingleSpeaker(cfg, dtype=dt).to(dev)
        st = m.store
        ff1 = m.text_encoder.transformer_blocks[0].ff.net[0].proj.weight
        assert id(ff1) not in st.geglu_ids and tuple(ff1.shape) == (512, 64)
        if dt == torch.bfloat16:          # the UNet's d = 256 projections keep the interleaved layout
            assert any(st.info[i]["name"].startswith("unet.") for i in st.geglu_ids)
        loss, gn = m.train_step(*batch)
        g_ff1 = st.grad_view(ff1).detach().float().cpu().clone()
        out[dt] = (float(loss), float(gn.sqrt()), g_ff1)
    l32, n32, g32 = out[torch.float32]; l16, n16, g16 = out[torch.bfloat16]
    assert abs(l16 - l32) < 2e-2 * abs(l32) and abs(n16 - n32) < 5e-2 * n32
    assert float((g16 - g32).norm() / g32.norm()) < 8e-2
