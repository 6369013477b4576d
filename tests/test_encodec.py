"""Encodec decode and encode: oracle vs the installed transformers EncodecModel (CPU), HIP decoder / encoder vs oracle (GPU)."""
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
@pytest.mark.parametrize("B,T,remote", [(5, 300, 0), (20, 150, 0), (70, 40, 0), (20, 150, 1), (70, 40, 1)])
def test_persistent_lstm_decoder_vs_oracle_and_step_kernels(dev, B, T, remote, monkeypatch):
    """bf16 decode at sizes that take the PERSISTENT LSTM (one launch for all T steps, weights resident in registers, hidden state
    exchanged between the cluster's 32 workgroups every step): 1, 3 and 8 + 1 clusters of 8 rows (70 rows = two launches),
    ragged last cluster; in the XCD-local form the census picks on a 256-CU device and (remote = 1) in the placement-independent
    form that exchanges through memory.  Checked against the CPU oracle (tolerance of test_hip_decoder_vs_oracle) and,
    tightly, against the same decoder in f32 mode."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    if remote:
        monkeypatch.setenv("PT_LSTM_FORCE_REMOTE", "1")
    W = oe.random_weights(5)
    codes = torch.randint(0, 1024, (B, 8, T), generator=torch.Generator().manual_seed(B + T))
    got = EncodecDecoder(W, device=dev, dtype=torch.bfloat16).decode(codes.to(dev)).cpu()
    f32 = EncodecDecoder(W, device=dev, dtype=torch.float32).decode(codes.to(dev)).cpu()
    assert torch.isfinite(got).all() and got.shape == (B, 1, 320 * T)
    assert float((got - f32).abs().max() / f32.abs().max()) < 6e-2
    rel_rms = float((got - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt())
    assert rel_rms < 1.5e-2, rel_rms                                   # a wrong hand-off corrupts whole rows, not the last bits
    want = oe.decode(codes[:2], W)
    assert float((got[:2] - want).abs().max() / want.abs().max()) < 6e-2


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(1, 7), (3, 33), (2, 101)])
def test_fused_24khz_tail_equals_the_separate_launches(dev, B, T, monkeypatch):
    """pt_encodec_tail (last transposed conv + residual block + final conv in one launch, intermediates in LDS) and
    pt_encodec_stage (the 3 kHz -> 12 kHz stage: transposed conv + residual block in one launch) and pt_encodec_res (the
    residual block of the 600 Hz -> 3 kHz stage, 62-row tiles + 2-row halo) against the separate launches they replace, same bf16 rounding points: item starts (reflect padding), tile seams (64-row
    tiles + 5-row halo; 33 and 101 frames are not multiples of anything) and item ends."""
    import prompt_tts_amd.encodec as pe
    from oracle import encodec as oe
    dec = pe.EncodecDecoder(oe.random_weights(8), device=dev, dtype=torch.bfloat16)
    codes = torch.randint(0, 1024, (B, 8, T), generator=torch.Generator().manual_seed(T)).to(dev)
    monkeypatch.setattr(pe, "FUSED_TAIL", True); monkeypatch.setattr(pe, "FUSED_STAGES", True)
    fused = dec.decode(codes).cpu()
    monkeypatch.setattr(pe, "FUSED_STAGES", False)
    tail_only = dec.decode(codes).cpu()
    monkeypatch.setattr(pe, "FUSED_TAIL", False)
    sep = dec.decode(codes).cpu()
    assert float((tail_only - sep).abs().max()) < 4e-3 * float(sep.abs().max())
    assert fused.shape == sep.shape == (B, 1, 320 * T)
    peak = float(sep.abs().max())
    # same bf16 rounding points, other summation orders inside the two fused kernels (bf16 eps = 3.9e-3 per rounding)
    assert float((fused - sep).abs().max()) < 1e-2 * peak, float((fused - sep).abs().max()) / peak
    assert float((fused - sep).pow(2).mean().sqrt()) < 2e-3 * peak
    want = oe.decode(codes.cpu(), oe.random_weights(8))
    assert float((fused - want).abs().max()) < 6e-2 * float(want.abs().max())


@pytest.mark.gpu
def test_decode_at_configs3_size_bf16_vs_f32_and_oracle(dev):
    """BASELINE configs[3]: 64 prompts x 1024 frames.  The f32-class decoder (the bench's headline path: split storage, fused
    f32-class stages, persistent LSTM over 1024 recurrent steps) on four full-length items against the CPU oracle at north_star's
    1e-3; the bf16 decoder against it on all 64 items (stated bound for the 1024-step bf16 recurrence: 8e-2 of the waveform peak,
    2e-2 relative RMS -- outside north_star's bound, which is why bf16 is not the headline)."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    W = oe.random_weights(6)
    codes = torch.randint(0, 1024, (64, 8, 1024), generator=torch.Generator().manual_seed(64))
    bf = EncodecDecoder(W, device=dev, dtype=torch.bfloat16).decode(codes.to(dev)).cpu()
    f32 = EncodecDecoder(W, device=dev, dtype=torch.float32).decode(codes.to(dev)).cpu()
    assert bf.shape == (64, 1, 327680) and torch.isfinite(bf).all()
    peak = float(f32.abs().max())
    assert float((bf - f32).abs().max()) < 8e-2 * peak
    assert float((bf - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()) < 2e-2
    for item in (0, 17, 40, 63):                                        # CPU oracle, four full-length items (first / last cluster rows too)
        want = oe.decode(codes[item:item + 1], W)
        err = float((f32[item:item + 1] - want).abs().max()) / float(want.abs().max())
        assert err < 1e-3, (item, err)                                  # north_star's bound, at the reference's precision (f32-class path)
        assert float((bf[item:item + 1] - want).abs().max()) < 8e-2 * float(want.abs().max())


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


# ---- encode (SURVEY a-12) ---------------------------------------------------------------------------------------------
def _hf_with_weights(W):
    """transformers EncodecModel carrying the effective encoder weights W (weight-norm parametrisation: v = w, g = |w|)."""
    from transformers import EncodecConfig, EncodecModel
    m = EncodecModel(EncodecConfig()).eval()

    def put(conv, w, b):
        with torch.no_grad():
            par = conv.parametrizations.weight
            par.original1.copy_(w); par.original0.copy_(w.flatten(1).norm(dim=1).view(-1, 1, 1)); conv.bias.copy_(b)
    Ls = m.encoder.layers
    put(Ls[0].conv, W["enc.conv0.w"], W["enc.conv0.b"])
    idx = 1
    for i in range(4):
        rb = Ls[idx]
        put(rb.block[1].conv, W[f"enc.res{i}.c3.w"], W[f"enc.res{i}.c3.b"])
        put(rb.block[3].conv, W[f"enc.res{i}.c1.w"], W[f"enc.res{i}.c1.b"])
        put(rb.shortcut.conv, W[f"enc.res{i}.sc.w"], W[f"enc.res{i}.sc.b"])
        put(Ls[idx + 2].conv, W[f"enc.down{i}.w"], W[f"enc.down{i}.b"])
        idx += 3
    lstm = Ls[idx].lstm
    with torch.no_grad():
        for l in range(2):
            for n in ("w_ih", "w_hh", "b_ih", "b_hh"):
                getattr(lstm, f"{'weight' if n[0] == 'w' else 'bias'}_{n[2:]}_l{l}").copy_(W[f"enc.lstm.{n}{l}"])
        for q in range(W["codebooks"].shape[0]):
            m.quantizer.layers[q].codebook.embed.copy_(W["codebooks"][q])
    put(Ls[idx + 2].conv, W["enc.final.w"], W["enc.final.b"])
    return m


def test_encode_oracle_matches_transformers_encodec_cpu():
    """The restated encoder + RVQ search against the independent implementation, with O(1) activations and varied codes."""
    from oracle import encodec as oe
    W = oe.random_encoder_weights(5)
    m = _hf_with_weights(W)
    wav = torch.randn(2, 1, 320 * 24, generator=torch.Generator().manual_seed(2)) * 0.5
    with torch.no_grad():
        want_emb = m.encoder(wav)
        want_codes = m.encode(wav, bandwidth=6.0).audio_codes[0]
    emb = oe.encoder_embeddings(wav, W)
    assert emb.shape == (2, 128, 24)
    assert float((emb - want_emb).abs().max()) < 1e-5 * float(want_emb.abs().max())
    codes = oe.encode(wav, W)
    assert codes.shape == (2, 8, 24) and codes.dtype == torch.int64
    assert torch.equal(codes, want_codes)
    assert codes.unique().numel() > 100                      # a real search, not a constant answer
    c64, gaps = oe.rvq_encode(emb, W["codebooks"], torch.float64)
    assert torch.equal(c64, codes) and float(gaps.min()) > 0
    with pytest.raises(ValueError):
        oe.encoder_embeddings(torch.zeros(1, 1, 321), W)


def test_encode_codec_tar_plumbing_cpu(tmp_path):
    """encode_codec.py keeps generate_code.py's on-disk format: <utt>.npy int64 [8,T], <utt>.len.txt = ceil(n/320), texts copied."""
    import io, tarfile, wave
    import numpy as np
    import encode_codec as ec
    tarp = str(tmp_path / "x.tar")
    rng = np.random.default_rng(0)
    with tarfile.open(tarp, "w") as tf:
        for i, n in enumerate((5000, 7777)):
            buf = io.BytesIO()
            with wave.open(buf, "wb") as w:
                w.setnchannels(2 if i else 1); w.setsampwidth(2); w.setframerate(24000)
                w.writeframes((rng.standard_normal(n * (2 if i else 1)) * 3000).astype("<i2").tobytes())
            for name, data in ((f"utt{i}.wav", buf.getvalue()), (f"utt{i}.txt", f"hello {i}".encode())):
                ti = tarfile.TarInfo(name); ti.size = len(data); tf.addfile(ti, io.BytesIO(data))

    class Fake:                                         # the tar / wav plumbing is host code; the encoder itself is GPU-only
        def encode(self, wav):
            assert wav.shape == (2, 1, 24000) and float(wav.abs().max()) < 1.0
            return torch.arange(wav.shape[0] * 8 * 75).view(wav.shape[0], 8, 75)
    old = ec._model
    ec._model = Fake()
    try:
        out = ec.main(tarp, 2, 1)
    finally:
        ec._model = old
    with tarfile.open(out) as tf:
        assert sorted(m.name for m in tf.getmembers()) == ["utt0.len.txt", "utt0.npy", "utt0.txt", "utt1.len.txt", "utt1.npy", "utt1.txt"]
        assert tf.extractfile("utt1.len.txt").read() == b"25.0" and tf.extractfile("utt0.txt").read() == b"hello 0"
        code = np.load(io.BytesIO(tf.extractfile("utt1.npy").read()))
        assert code.dtype == np.int64 and code.shape == (8, 75) and code[0, 0] == 600
    # other sample rates are resampled to 24 kHz (generate_code.py:28 convert_audio): length ceil(n * 24000 / rate); a tone stays
    # the same tone (band-limited sinc interpolation, torchaudio Resample defaults), 24-bit PCM is read too
    import math
    for rate, width in ((16000, 2), (44100, 3), (24000, 2)):
        n = 3000
        tone = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(n) / rate)
        buf = io.BytesIO()
        with wave.open(buf, "wb") as w:
            w.setnchannels(1); w.setsampwidth(width); w.setframerate(rate)
            if width == 2:
                w.writeframes((tone * 32767).astype("<i2").tobytes())
            else:
                v = (tone * 8388607).astype(np.int32)
                w.writeframes(np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], 1).astype(np.uint8).tobytes())
        buf.seek(0)
        got = ec.read_wav(buf)
        m = int(math.ceil(n * 24000 / rate))
        assert got.shape == (1, m) and got.dtype == torch.float32
        want = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(m) / 24000.0)
        inner = slice(64, m - 64)                                        # away from the zero-padded edges of the finite signal
        assert float(np.abs(got[0].numpy()[inner] - want[inner]).max()) < 2e-3
    assert torch.equal(ec.resample(torch.ones(2, 10), 8000, 8000), torch.ones(2, 10))
    # no weights loaded: generate() / decode() refuse instead of silently using a random codec
    ec._model = None
    with pytest.raises(RuntimeError):
        ec.generate([torch.zeros(1, 1, 320)])
    with pytest.raises(RuntimeError):
        ec.load_encoder(None)
    import decode_codec as dc
    dc._model = None
    with pytest.raises(RuntimeError):
        dc.decode(torch.zeros(1, 8, 10, dtype=torch.long))
    with pytest.raises(BaseException):
        dc.decode(torch.zeros(8, 10, dtype=torch.long))                 # the reference's shape check comes first


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(2, 24), (3, 50)])
def test_hip_encoder_vs_oracle(dev, B, T):
    """f32: embeddings within 1e-3; the search is bit-exact given its input (codes == f64 search of the device embeddings,
    except where the f64 runner-up is within 1e-4 of the winner); end-to-end code agreement with the oracle reported."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecEncoder
    W = oe.random_encoder_weights(5)
    wav = torch.randn(B, 1, 320 * T, generator=torch.Generator().manual_seed(10 * B + T)) * 0.5
    enc = EncodecEncoder(W, device=dev, dtype=torch.float32)
    emb, b, t = enc.embeddings(wav.to(dev))
    want = oe.encoder_embeddings(wav, W)                                   # (B,128,T)
    got = emb.view(B, T, 128).permute(0, 2, 1).cpu()
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-3, err
    codes = enc.quantize(emb, b, t).cpu()
    assert codes.shape == (B, 8, T) and codes.dtype == torch.int64
    ref, gaps = oe.rvq_encode(got, W["codebooks"], torch.float64)          # same input, reference arithmetic
    # a token's later stages depend on its earlier choices: compare up to and including the first disagreement
    bad = codes != ref
    first_bad = bad.int().argmax(dim=1, keepdim=True)                      # (B,1,T) stage of the first disagreement (0 if none)
    any_bad = bad.any(dim=1, keepdim=True)
    g_at = gaps.gather(1, first_bad)
    assert bool(((~any_bad) | (g_at < 1e-4)).all()), "search disagrees away from a near-tie"
    assert float(any_bad.float().mean()) < 0.01
    e2e = oe.encode(wav, W)
    assert float((codes == e2e).float().mean()) > 0.97                     # embeddings differ by ~1e-6: only near-ties flip


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,rows16", [(20, 60, 0), (70, 24, 0), (20, 60, 1)])
def test_exact_f32_persistent_lstm_of_the_encoder(dev, B, T, rows16, monkeypatch):
    """The encoder's LSTM at sizes that take the PERSISTENT exact-f32 form (v_mfma_f32_16x16x4_f32, f32 hidden values in the
    granules; 2 clusters; 4 + 1 clusters in two launches): embeddings against the per-step kernels (same arithmetic, another
    summation order: 1e-5 of the peak) and against the CPU oracle (1e-3), codes against the oracle's."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecEncoder
    W = oe.random_encoder_weights(6)
    wav = torch.randn(B, 1, 320 * T, generator=torch.Generator().manual_seed(B + T)) * 0.5
    enc = EncodecEncoder(W, device=dev, dtype=torch.float32)
    if rows16:                                                            # the 16-row x 64-workgroup kernel instead of 8 rows per XCD
        monkeypatch.setenv("PT_LSTM_F32_ROWS8", "0")
    emb, b, t = enc.embeddings(wav.to(dev))
    monkeypatch.setenv("PT_LSTM_PERSIST_EXACT", "0")
    emb_step, _, _ = enc.embeddings(wav.to(dev))
    monkeypatch.delenv("PT_LSTM_PERSIST_EXACT")
    peak = float(emb_step.abs().max())
    assert float((emb - emb_step).abs().max()) < 1e-5 * peak, float((emb - emb_step).abs().max()) / peak
    want = oe.encoder_embeddings(wav[:3], W)
    got = emb.view(B, T, 128)[:3].permute(0, 2, 1).cpu()
    assert float((got - want).abs().max() / want.abs().max()) < 1e-3
    codes = enc.quantize(emb, b, t).cpu()
    assert float((codes[:3] == oe.encode(wav[:3], W)).float().mean()) > 0.97


@pytest.mark.gpu
def test_hip_encoder_at_generate_code_size(dev):
    """data_preparation/generate_code.py:96 encodes batches of 32 windows x 12 s (288 000 samples -> 900 frames).  The HIP encoder
    at exactly that size (two persistent-LSTM launches: 32 rows = 4 clusters... of 8), three of the 32 items against the CPU oracle:
    embeddings within 1e-3 of the peak, the search bit-exact given its input (codes == f64 search of the device embeddings except
    at f64 near-ties < 1e-4), end-to-end code agreement >= 97 % (embeddings differ at ~1e-6, only near-ties flip)."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecEncoder
    W = oe.random_encoder_weights(5)
    B, T = 32, 900
    wav = torch.randn(B, 1, 320 * T, generator=torch.Generator().manual_seed(96)) * 0.5
    enc = EncodecEncoder(W, device=dev, dtype=torch.float32)
    emb, b, t = enc.embeddings(wav.to(dev))
    assert (b, t) == (B, T)
    codes = enc.quantize(emb, b, t).cpu()
    assert codes.shape == (B, 8, T) and codes.dtype == torch.int64 and int(codes.min()) >= 0 and int(codes.max()) < 1024
    got_all = emb.view(B, T, 128).permute(0, 2, 1).cpu()
    for item in (0, 13, 31):
        want = oe.encoder_embeddings(wav[item:item + 1], W)
        got = got_all[item:item + 1]
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 1e-3, (item, err)
        ref, gaps = oe.rvq_encode(got, W["codebooks"], torch.float64)
        bad = codes[item:item + 1] != ref
        first_bad = bad.int().argmax(dim=1, keepdim=True)
        any_bad = bad.any(dim=1, keepdim=True)
        assert bool(((~any_bad) | (gaps.gather(1, first_bad) < 1e-4)).all()), "search disagrees away from a near-tie"
        assert float((codes[item:item + 1] == oe.encode(wav[item:item + 1], W)).float().mean()) > 0.97


@pytest.mark.gpu
def test_hip_encoder_causal_and_batch_independent(dev):
    """Size-independent properties at a longer length: items are independent; a prefix encodes to the prefix of the codes."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecEncoder
    enc = EncodecEncoder(oe.random_encoder_weights(6), device=dev, dtype=torch.float32)
    wav = (torch.randn(3, 1, 320 * 150, generator=torch.Generator().manual_seed(3)) * 0.5).to(dev)
    full = enc.encode(wav)
    assert torch.equal(enc.encode(wav[1:2]), full[1:2])
    assert torch.equal(enc.encode(wav[:, :, :320 * 90]), full[:, :, :90])
    with pytest.raises(ValueError):
        enc.encode(wav[:, :, :321])


@pytest.mark.gpu
def test_hip_encoder_bf16_embeddings(dev):
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecEncoder
    W = oe.random_encoder_weights(5)
    wav = torch.randn(2, 1, 320 * 30, generator=torch.Generator().manual_seed(4)) * 0.5
    emb, B, T = EncodecEncoder(W, device=dev, dtype=torch.bfloat16).embeddings(wav.to(dev))
    want = oe.encoder_embeddings(wav, W)
    err = float((emb.view(B, T, 128).permute(0, 2, 1).cpu() - want).abs().max() / want.abs().max())
    assert err < 8e-2, err


@pytest.mark.gpu
def test_persistent_lstm_timeout_raises_instead_of_returning_garbage(dev, monkeypatch):
    """VERDICT r02 / ADVICE r02: a lost hand-off of the persistent LSTM (its workgroups not all resident) used to return a garbage
    waveform with PT_OK.  Force the timeout branch -- the workgroup in role (cluster 0, slice 5) publishes wrong tags, the spin bound is shrunk so that the
    launch gives up after a few milliseconds.  With PT_LSTM_RETRY=0 decode() must RAISE; by default it retries the call once with
    the per-step kernels (pt_lstm2_desc.per_step; ADVICE r03: a running process could not switch forms) and returns a waveform
    that matches the good one to the rounding of another summation order; then decode again without the fault: the status word is
    per call and the result is bit-identical to the first."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    W = oe.random_weights(5)
    dec = EncodecDecoder(W, device=dev, dtype=torch.bfloat16)
    codes = torch.randint(0, 1024, (20, 8, 150), generator=torch.Generator().manual_seed(3)).to(dev)   # 3 clusters, persistent form
    good = dec.decode(codes).cpu()
    monkeypatch.setenv("PT_LSTM_DEBUG_SPIN", "2000")
    monkeypatch.setenv("PT_LSTM_DEBUG_FAULT_SLICE", "5")
    monkeypatch.setenv("PT_LSTM_RETRY", "0")
    with pytest.raises(RuntimeError, match="timed out"):
        dec.decode(codes)
    torch.cuda.synchronize()
    monkeypatch.delenv("PT_LSTM_RETRY")
    retried = dec.decode(codes).cpu()                                 # timed out, then the per-step kernels
    assert torch.isfinite(retried).all()
    assert float((retried - good).abs().max()) < 3e-2 * float(good.abs().max())
    monkeypatch.delenv("PT_LSTM_DEBUG_SPIN"); monkeypatch.delenv("PT_LSTM_DEBUG_FAULT_SLICE")
    again = dec.decode(codes).cpu()
    assert torch.equal(again, good)


@pytest.mark.gpu
def test_persistent_lstm_timeout_in_a_later_launch_of_the_call_is_sticky(dev, monkeypatch):
    """70 rows = two launches (8 clusters of 8 rows + 1): the fault sits in (cluster 2, slice 6) = role 70, which exists only in
    the FIRST launch; the second launch must not clear the word."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    dec = EncodecDecoder(oe.random_weights(5), device=dev, dtype=torch.bfloat16)
    codes = torch.randint(0, 1024, (70, 8, 40), generator=torch.Generator().manual_seed(4)).to(dev)
    monkeypatch.setenv("PT_LSTM_DEBUG_SPIN", "2000")
    monkeypatch.setenv("PT_LSTM_DEBUG_FAULT_SLICE", "70")
    monkeypatch.setenv("PT_LSTM_RETRY", "0")
    with pytest.raises(RuntimeError, match="timed out"):
        dec.decode(codes)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,form", [(20, 150, "local"), (70, 40, "local"), (5, 300, "local"), (20, 150, "remote"), (70, 40, "rows16")])
def test_persistent_f32_lstm_decoder_vs_oracle(dev, B, T, form, monkeypatch):
    """f32 decode at sizes that take the PERSISTENT f32-class LSTM (bf16 x 3 products, hi / lo weight fragments in registers / LDS,
    16-byte data-tagged granules): 3, 8 + 1 (two launches, ragged last cluster) and 1 cluster(s) of 8 rows on one XCD each
    ("local"), the same kernel exchanging through memory ("remote"), and the 16-row x 64-workgroup kernel ("rows16"), against the
    CPU oracle at the north_star bound for floating point (1e-3 of the waveform peak; the exact-f32 per-step kernels: ~1e-6)."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    if form == "remote":
        monkeypatch.setenv("PT_LSTM_FORCE_REMOTE", "1")
    if form == "rows16":
        monkeypatch.setenv("PT_LSTM_F32_ROWS8", "0")
    W = oe.random_weights(5)
    codes = torch.randint(0, 1024, (B, 8, T), generator=torch.Generator().manual_seed(B + T))
    got = EncodecDecoder(W, device=dev, dtype=torch.float32).decode(codes.to(dev)).cpu()
    want = oe.decode(codes, W)
    assert got.shape == want.shape == (B, 1, 320 * T) and torch.isfinite(got).all()
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-3, err


@pytest.mark.gpu
@pytest.mark.parametrize("half", [0, 1, 2])
def test_persistent_f32_lstm_timeout_raises(dev, half, monkeypatch):
    """the 16-byte granules of the f32-class / exact-f32 kernels carry one tag PER 8-BYTE HALF ({data, tag, data, tag});
    a granule whose first (half = 1) or second (half = 2) half alone is stale -- what a 16-byte access split at the 8-byte boundary
    would show -- must fail the consumers' check exactly like one with both tags wrong (half = 0).  By default the call is then
    retried with the per-step kernels and matches the oracle."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    W = oe.random_weights(5)
    dec = EncodecDecoder(W, device=dev, dtype=torch.float32)
    codes = torch.randint(0, 1024, (20, 8, 150), generator=torch.Generator().manual_seed(3)).to(dev)
    monkeypatch.setenv("PT_LSTM_DEBUG_SPIN", "2000")
    monkeypatch.setenv("PT_LSTM_DEBUG_FAULT_SLICE", "5")
    monkeypatch.setenv("PT_LSTM_DEBUG_FAULT_HALF", str(half))
    monkeypatch.setenv("PT_LSTM_RETRY", "0")
    with pytest.raises(RuntimeError, match="timed out"):
        dec.decode(codes)
    torch.cuda.synchronize()
    if half == 1:
        monkeypatch.delenv("PT_LSTM_RETRY")
        got = dec.decode(codes).cpu()
        want = oe.decode(codes[:2].cpu(), W)
        assert float((got[:2] - want).abs().max()) < 1e-3 * float(want.abs().max())


@pytest.mark.gpu
def test_two_decoders_on_two_streams_equal_the_sequential_decodes(dev):
    """VERDICT r03 item 2 (the run that wrote nothing for 420 s, and three overlapping decodes that returned DIFFERENT waveforms
    without an error): two decoders -- and two calls on ONE decoder -- enqueued on two streams at once.  The library orders the
    persistent LSTM launches of a device itself and every call owns its scratch and status word, so the results must be the
    bits of the sequential decodes.  Run once, small T."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    W = oe.random_weights(5)
    d1 = EncodecDecoder(W, device=dev, dtype=torch.bfloat16)
    d2 = EncodecDecoder(W, device=dev, dtype=torch.float32)
    c1 = torch.randint(0, 1024, (20, 8, 150), generator=torch.Generator().manual_seed(1)).to(dev)
    c2 = torch.randint(0, 1024, (24, 8, 140), generator=torch.Generator().manual_seed(2)).to(dev)
    want1, want2, want3 = d1.decode(c1).clone(), d2.decode(c2).clone(), d1.decode(c2).clone()
    torch.cuda.synchronize()
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    outs = {}
    import threading

    def work(key, dec, codes, stream):
        with torch.cuda.stream(stream):
            outs[key] = dec.decode(codes)
        stream.synchronize()
    th = [threading.Thread(target=work, args=a) for a in (("a", d1, c1, s1), ("b", d2, c2, s2), ("c", d1, c2, s3))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th)
    torch.cuda.synchronize()
    assert torch.equal(outs["a"], want1) and torch.equal(outs["b"], want2) and torch.equal(outs["c"], want3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,n", [(torch.bfloat16, 12), (torch.float32, 5)])
def test_decode_is_bit_identical_from_launch_to_launch(dev, dtype, n):
    """configs[3] size (64 x 1024 frames, 1024 recurrent ticks, 8 XCD-local clusters): the census hands out the unit slices of a
    cluster in arrival order, so WHICH workgroup computes which hidden units changes from launch to launch -- the arithmetic
    never does.  Every decode must return the first one's waveform bit for bit (a stale or torn hand-off would not), and no
    status word may fire (decode() raises)."""
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    dec = EncodecDecoder(oe.random_weights(6), device=dev, dtype=dtype)
    codes = torch.randint(0, 1024, (64, 8, 1024), generator=torch.Generator().manual_seed(11)).to(dev)
    ref = dec.decode(codes).clone()
    assert torch.isfinite(ref).all()
    for _ in range(n):
        assert torch.equal(dec.decode(codes), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,x3", [(torch.float32, 1), (torch.float32, 0), (torch.bfloat16, 0)])
@pytest.mark.parametrize("cin,taps,N,rowmap,stride", [(32, 7, 1, "causal", 1), (64, 3, 32, "causal", 1), (64, 2, 64, "back", 1),
                                                      (32, 4, 64, "strided", 2), (16, 3, 16, "causal", 1)])
def test_rowconv_staged_through_lds_equals_the_global_memory_kernel(dev, dtype, x3, cin, taps, N, rowmap, stride, monkeypatch):
    """pt_rowconv with its input rows staged through LDS (interior blocks; input ELU and hi / lo split applied while staging)
    against the kernel that fetches every tap from global memory (PT_ROWCONV_STAGED=0): same products in the same order -> the
    same bits in the interior; three items of 2 024 rows so that blocks at item starts / ends (reflect, zero padding, two items in one block)
    take the edge path inside the staged kernel."""
    import ctypes as C
    from prompt_tts_amd import _lib as L, ops
    from prompt_tts_amd._lib import lib
    g = torch.Generator().manual_seed(cin * 100 + taps)
    B, n_out = 3, 2000 + 24                     # M = 6 072 rows: above the staged kernel's 4 096-row threshold
    n_in = n_out * stride
    K = taps * cin
    ldw = (K + 31) // 32 * 32
    x = torch.randn(B * n_in, cin, generator=g).to(dev, dtype)
    w = torch.zeros(N, ldw); w[:, :K] = torch.randn(N, K, generator=g) * K ** -0.5
    w = w.to(dev, dtype)
    bias = torch.randn(N, generator=g).to(dev)
    rm = {"causal": L.PT_MAP_CAUSAL_REFLECT, "back": L.PT_MAP_BACK, "strided": L.PT_MAP_STRIDED_REFLECT}[rowmap]

    def run():
        y = torch.full((B * n_out, max(N, 4)), float("nan"), device=dev, dtype=dtype)
        d = L.pt_rowconv_desc()
        d.B, d.n_rows = B, n_out
        d.x, d.ldx, d.cin, d.taps, d.rowmap, d.elu_x, d.stride = x.data_ptr(), x.stride(0), cin, taps, rm, 1, stride
        d.w, d.ldw, d.bias, d.N, d.act = w.data_ptr(), w.stride(0), bias.data_ptr(), N, 1
        d.y, d.ldy, d.y_f32, d.f32_x3 = y.data_ptr(), y.stride(0), 0, x3
        ops.check(lib.pt_rowconv(C.byref(d), ops._DT[dtype], ops._stream()), "pt_rowconv")
        return y[:, :N].clone()
    staged = run()
    monkeypatch.setenv("PT_ROWCONV_STAGED", "0")
    plain = run()
    assert torch.isfinite(staged).all()
    s3, p3 = staged.view(B, n_out, N), plain.view(B, n_out, N)
    assert torch.equal(s3[:, 128:n_out - 128], p3[:, 128:n_out - 128])          # blocks staged through LDS: the same bits
    # blocks at an item edge add their k-steps in another order (last-bit differences)
    assert float((s3.float() - p3.float()).abs().max()) <= 4e-6 * float(p3.float().abs().max())
    # and against plain torch on a few rows of the middle item (f32 exact only)
    if dtype == torch.float32 and not x3 and rowmap == "causal":
        xe = torch.nn.functional.elu(x.float().cpu().view(B, n_in, cin))[1]
        t = 500
        ref = sum(xe[t + tap - (taps - 1)] @ w.float().cpu()[:, tap * cin:(tap + 1) * cin].T for tap in range(taps)) + bias.cpu()
        ref = torch.nn.functional.elu(ref)
        assert float((staged[n_out + t].float().cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
