"""f32-class decode on split storage (PT_BF16X2; csrc/encodec_x2.hip, pt_gemm's plane operands): the reference decodes in fp32
(decode_codec.py:12-16) and north_star's bound for floating point is 1e-3 relative.  Pieces against plain torch f32, the whole
decoder against the CPU oracle and against round 3's f32 path (hi / lo split inside the product loops)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _planes(x):
    from prompt_tts_amd.encodec import planes_from_f32
    return planes_from_f32(x)


def _unplanes(x, block=None):
    """(rows, 2N) plane rows -> (rows, N) f32; block = plane blocking of the columns (None: one block)."""
    rows, two_n = x.shape
    n = two_n // 2
    P = block or n
    v = x.float().view(rows, n // P, 2, P)
    return (v[:, :, 0] + v[:, :, 1]).reshape(rows, n)


def test_planes_round_trip_is_f32_class(dev):
    x = torch.randn(37, 64, generator=torch.Generator().manual_seed(0)) * 3
    back = _unplanes(_planes(x))
    assert float((back - x).abs().max()) < 2.0 ** -15 * float(x.abs().max())


@pytest.mark.parametrize("M,N,K,kind", [(300, 128, 192, "plain"), (1024, 256, 512, "plain"), (225, 64, 384, "concat"),
                                        (3 * 75, 512, 7 * 128, "conv7"), (2 * 300, 640, 2 * 256, "back2"), (2 * 96, 128, 3 * 256, "conv3"),
                                        (1, 8, 32, "plain"), (257, 72, 32, "plain"), (130, 136, 160, "concat"), (3 * 11, 24, 7 * 32, "conv7"),
                                        (2 * 5, 128, 2 * 32, "back2"), (2 * 700, 256, 3 * 32, "conv3")])
def test_gemm_x2_matches_f32(dev, M, N, K, kind):
    """pt_gemm(PT_BF16X2): plane operands (plain, channel concat, causal-reflect / back conv gathers), bias + ELU, a second ELU'd
    output, plane-blocked output columns, f32 output; M tails and N that is not a multiple of a tile; one k-tile (K = 32), a
    32-column segment per conv tap / concat half, items shorter than the taps reach -- against torch f32 at 1e-4 of the output's
    peak (three bf16 products carry ~2^-16 per product)."""
    from prompt_tts_amd import _lib as L, ops
    g = torch.Generator().manual_seed(M + N + K)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    bias = torch.randn(N, generator=g)
    if kind == "plain":
        x = torch.randn(M, K, generator=g)
        A = lambda xd: ops.plain(xd)
        ref_in = x
    elif kind == "concat":
        x1 = torch.randn(M, 128, generator=g); x2 = torch.randn(M, K - 128, generator=g)
        ref_in = torch.cat([x1, x2], 1)
    else:
        taps = {"conv7": 7, "back2": 2, "conv3": 3}[kind]
        cin = K // taps
        Bn = {"conv7": 3, "back2": 2, "conv3": 2}[kind]
        n = M // Bn
        x = torch.randn(Bn * n, cin, generator=g)
        xi = x.view(Bn, n, cin)
        cols = []
        for tap in range(taps):
            if kind == "back2":
                src = torch.arange(n) - tap
                part = torch.where((src >= 0)[None, :, None], xi[:, src.clamp(min=0)], torch.zeros(()))
            else:
                src = (torch.arange(n) + tap - (taps - 1)).abs()
                part = xi[:, src]
            cols.append(part)
        ref_in = torch.cat(cols, dim=2).reshape(M, K)
    want = ref_in.double() @ w.double().T + bias.double()
    wp = _planes(w).to(dev)
    bd = bias.to(dev)
    if kind == "plain":
        xd = _planes(x).to(dev); Aop = ops.plain(xd)
    elif kind == "concat":
        a, b = _planes(x1).to(dev), _planes(x2).to(dev)
        Aop = ops.concat(a, b); Aop.c_split = 128
    else:
        xd = _planes(x).to(dev)
        rm = L.PT_MAP_BACK if kind == "back2" else L.PT_MAP_CAUSAL_REFLECT
        Aop = ops.conv(xd, cin, n, n, rm, taps=taps)
    block = {640: 128, 512: 256}.get(N, 0)
    out = torch.full((M, 2 * N), float("nan"), device=dev, dtype=torch.bfloat16)
    out2 = torch.full((M, 2 * N), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.gemm(M, N, K, Aop, ops.plain(wp), out, L.PT_BF16X2, ldc=2 * N, bias=bd, x2_block=block, out2=out2, ldc2=2 * N, act2=1)
    of = torch.empty(M, N, device=dev, dtype=torch.float32)
    ops.gemm(M, N, K, Aop, ops.plain(wp), of, L.PT_BF16X2, out_kind=L.PT_OUT_F32, bias=bd)
    one = torch.full((M, 2 * N), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.gemm(M, N, K, Aop, ops.plain(wp), one, L.PT_BF16X2, ldc=2 * N, bias=bd, act=1)
    torch.cuda.synchronize()
    peak = float(want.abs().max())
    got = _unplanes(out.cpu(), block or None)
    assert float((got.double() - want).abs().max()) < 1e-4 * peak, float((got.double() - want).abs().max()) / peak
    assert float((of.cpu().double() - want).abs().max()) < 1e-4 * peak
    elu = torch.nn.functional.elu(want)
    assert float((_unplanes(out2.cpu(), block or None).double() - elu).abs().max()) < 1e-4 * peak
    assert float((_unplanes(one.cpu()).double() - elu).abs().max()) < 1e-4 * peak


def test_rvq_decode_x2(dev):
    from prompt_tts_amd import _lib as L, ops
    from prompt_tts_amd._lib import lib
    g = torch.Generator().manual_seed(1)
    cb = torch.randn(8, 1024, 128, generator=g)
    codes = torch.randint(0, 1024, (3, 8, 50), generator=g)
    want = sum(cb[q][codes[:, q]] for q in range(8)).reshape(150, 128)
    out = torch.empty(150, 256, device=dev, dtype=torch.bfloat16)
    cbd, cd = cb.to(dev), codes.to(dev)
    ops.check(lib.pt_rvq_decode(cd.data_ptr(), cbd.data_ptr(), out.data_ptr(), 3, 8, 50, 1024, 128, L.PT_BF16X2, ops._stream()), "rvq")
    assert float((_unplanes(out.cpu()) - want).abs().max()) < 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("B,T", [(2, 24), (3, 75), (1, 7), (5, 130), (2, 257)])
def test_x2_decoder_vs_oracle_and_vs_the_in_loop_split_path(dev, B, T, monkeypatch):
    """Whole decoder on split storage: fused f32-class res / stage / tail kernels (tile seams: 62-, 30- and 56-row tiles + halos;
    T = 7, 24, 75, 130, 257 put item starts, ragged last tiles and several tiles per item in play), plane GEMMs, the LSTM through
    its small-input fallback (B T < 513 rows) and -- (5, 130), (2, 257) -- the persistent plane form.  1e-3 of the waveform peak
    vs the CPU oracle (north_star), 2e-4 vs round 3's f32 path."""
    import prompt_tts_amd.encodec as pe
    from oracle import encodec as oe
    W = oe.random_weights(3)
    codes = torch.randint(0, 1024, (B, 8, T), generator=torch.Generator().manual_seed(B * 100 + T))
    want = oe.decode(codes, W)
    dec = pe.EncodecDecoder(W, device=dev, dtype=torch.float32)
    assert dec.x2
    got = dec.decode(codes.to(dev)).cpu()
    assert got.shape == (B, 1, 320 * T) and got.dtype == torch.float32 and torch.isfinite(got).all()
    peak = float(want.abs().max())
    err = float((got - want).abs().max()) / peak
    assert err < 1e-3, err
    monkeypatch.setattr(pe, "F32_PLANES", False)
    old = pe.EncodecDecoder(W, device=dev, dtype=torch.float32)
    assert not old.x2
    ref = old.decode(codes.to(dev)).cpu()
    assert float((got - ref).abs().max()) < 2e-4 * peak, float((got - ref).abs().max()) / peak


def test_x2_decoder_items_are_independent_and_causal(dev):
    from oracle import encodec as oe
    from prompt_tts_amd.encodec import EncodecDecoder
    dec = EncodecDecoder(oe.random_weights(4), device=dev, dtype=torch.float32)
    codes = torch.randint(0, 1024, (4, 8, 200), generator=torch.Generator().manual_seed(1)).to(dev)
    full = dec.decode(codes)
    one = dec.decode(codes[2:3])
    assert float((full[2:3] - one).abs().max()) < 1e-4 * float(full.abs().max())
    pre = dec.decode(codes[:, :, :120])
    assert float((full[:, :, :320 * 120] - pre).abs().max()) < 1e-4 * float(full.abs().max())
