"""Encodec decode: oracle vs the installed transformers EncodecModel (CPU), HIP decoder vs oracle (GPU)."""
import pytest
import torch


def test_oracle_matches_transformers_encodec_cpu():
    """Independent implementation of the same architecture (random weights; pretrained ones are fetched by URL upstream)."""
    from transformers import EncodecConfig, EncodecModel
    from oracle import encodec as oe
    torch.manual_seed(0)
    m = EncodecModel(EncodecConfig()).eval()
    for q in m.quantizer.layers:
        q.codebook.embed.normal_(0, 0.5)
    W = oe.weights_from_hf(m)
    for B, T in ((2, 37), (1, 7)):
        codes = torch.randint(0, 1024, (B, 8, T))
        with torch.no_grad():
            want = m.decode(codes[None], [None])[0]
            got = oe.decode(codes, W)
        assert got.shape == (B, 1, 320 * T)
        assert float((want - got).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    with pytest.raises(BaseException):
        oe.decode(torch.zeros(8, 10, dtype=torch.long), W)


def test_weight_norm_fold_and_state_dict_mapping_cpu():
    from prompt_tts_amd.encodec import fold_weight_norm, weights_from_encodec_state_dict, WEIGHT_KEYS
    v = torch.randn(6, 4, 3); g = torch.rand(6, 1, 1) + 0.5
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(4, 6, 3))
    with torch.no_grad():
        conv.weight_v.copy_(v); conv.weight_g.copy_(g)
    conv(torch.zeros(1, 4, 8))
    assert torch.allclose(fold_weight_norm(g, v), conv.weight, atol=1e-6)
    # synthetic state_dict in the original encodec package's naming
    sd = {}
    def add(prefix, shape, gdim0):
        sd[prefix + ".weight_v"] = torch.randn(shape); sd[prefix + ".weight_g"] = torch.rand(gdim0, 1, 1) + 0.5
        sd[prefix + ".bias"] = torch.randn(shape[0] if "convtr" not in prefix else shape[1])
    for q in range(8):
        sd[f"quantizer.vq.layers.{q}._codebook.embed"] = torch.randn(1024, 128)
    add("decoder.model.0.conv.conv", (512, 128, 7), 512)
    for l in range(2):
        for k, shp in (("weight_ih", (2048, 512)), ("weight_hh", (2048, 512)), ("bias_ih", (2048,)), ("bias_hh", (2048,))):
            sd[f"decoder.model.1.lstm.{k}_l{l}"] = torch.randn(shp)
    Cc, idx = 512, 3
    for r in (8, 5, 4, 2):
        add(f"decoder.model.{idx}.convtr.convtr", (Cc, Cc // 2, 2 * r), Cc)
        Cc //= 2
        add(f"decoder.model.{idx + 1}.block.1.conv.conv", (Cc // 2, Cc, 3), Cc // 2)
        add(f"decoder.model.{idx + 1}.block.3.conv.conv", (Cc, Cc // 2, 1), Cc)
        add(f"decoder.model.{idx + 1}.shortcut.conv.conv", (Cc, Cc, 1), Cc)
        idx += 3
    add(f"decoder.model.{idx}.conv.conv", (1, 32, 7), 1)
    W = weights_from_encodec_state_dict(sd)
    assert set(W) == set(WEIGHT_KEYS) and W["up0.w"].shape == (512, 256, 16) and W["final.w"].shape == (1, 32, 7)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("B,T", [(2, 24), (3, 75), (1, 7)])
def test_hip_decoder_vs_oracle(dev, dtype, tol, B, T):
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    W = oe.random_weights(3)
    codes = torch.randint(0, 1024, (B, 8, T), generator=torch.Generator().manual_seed(B * 100 + T))
    want = oe.decode(codes, W)
    dec = EncodecDecoder(W, device=dev, dtype=dtype)
    got = dec.decode(codes.to(dev))
    assert got.shape == (B, 1, 320 * T) and got.dtype == torch.float32
    err = float((got.cpu() - want).abs().max() / want.abs().max())
    assert err < tol, err
    with pytest.raises(BaseException):
        dec.decode(codes[0].to(dev))


@pytest.mark.gpu
def test_hip_decoder_batch_items_independent(dev):
    """Full-size property (size independent): decoding a batch equals decoding its items one by one; causal in time."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    dec = EncodecDecoder(oe.random_weights(4), device=dev, dtype=torch.float32)
    codes = torch.randint(0, 1024, (4, 8, 200), generator=torch.Generator().manual_seed(1)).to(dev)
    full = dec.decode(codes)
    one = dec.decode(codes[2:3])
    assert float((full[2:3] - one).abs().max()) < 1e-4 * float(full.abs().max())
    pre = dec.decode(codes[:, :, :120])                       # causal: a prefix decodes to the prefix of the waveform
    assert float((full[:, :, :320 * 120] - pre).abs().max()) < 1e-4 * float(full.abs().max())
