"""GPU: fp8 path of BASELINE configs[4] -- per-tensor quantisation and the fp8 MFMA GEMM against torch's own float8 dtypes.

The quantiser is checked BYTE-exact against torch (same amax, same scale arithmetic, round-to-nearest-even casts to
torch.float8_e4m3fn / float8_e5m2); the GEMM (K < 3072: two-stage 256 x 256 kernel; K >= 3072: eight-phase kernel) against an f32 matmul of the SAME fp8 operands (so only the f32 summation order and
the bf16 rounding of the output differ: one bf16 ulp per element), and against the unquantised product with the
tolerance fp8 itself allows (3 mantissa bits: 2^-4 per element, averaged over K)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

F8 = {0: (torch.float8_e4m3fn, 448.0), 1: (torch.float8_e5m2, 57344.0)}


def _ref_quant(x, fmt):
    dt, fmax = F8[fmt]
    amax = x.float().abs().max()
    scale = amax / fmax if float(amax) > 0 else torch.ones((), device=x.device)
    inv = fmax / amax if float(amax) > 0 else torch.ones((), device=x.device)
    q = (x.float() * inv).clamp(-fmax, fmax).to(dt)
    return q, float(amax), float(scale)


def _assert_bf16_close(got, want):
    """one bf16 rounding of the output (half an ulp = 2^-9 relative) + f32 summation-order noise."""
    err = (got.double() - want).abs()
    bound = 2.0 ** -8 * want.abs() + 1e-3 * float(want.pow(2).mean().sqrt())
    assert bool((err <= bound).all()), float((err - bound).max())


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("rows,cols,spread", [(256, 1024, 1.0), (1000, 80, 6.0), (64, 64, 0.0)])
def test_quantize_bytes_equal_torch_float8(dev, fmt, rows, cols, spread):
    from prompt_tts_amd import ops
    g = torch.Generator().manual_seed(rows + cols + fmt)
    x = (torch.randn(rows, cols, generator=g) * torch.exp(spread * torch.randn(rows, 1, generator=g))).to(torch.bfloat16).to(dev)
    if spread == 0.0:
        x.zero_()                                                     # all-zero tensor: scale 1, zeros out
    out = torch.empty(rows, cols, dtype=torch.uint8, device=dev)
    state = torch.empty(258, dtype=torch.float32, device=dev)
    out_t = torch.empty(cols, rows, dtype=torch.uint8, device=dev) if rows % 64 == 0 and cols % 64 == 0 else None
    ops.fp8_quantize(x, out, state, fmt, out_t=out_t)
    q, amax, scale = _ref_quant(x, fmt)
    assert float(state[0]) == amax and float(state[1]) == pytest.approx(scale, rel=1e-6)
    assert torch.equal(out, q.view(torch.uint8))
    if out_t is not None:
        assert torch.equal(out_t, q.view(torch.uint8).t().contiguous())


@pytest.mark.parametrize("a_fmt", [0, 1])
@pytest.mark.parametrize("M,N,K", [(512, 512, 1024), (768, 1280, 256), (300, 520, 144), (256, 256, 128), (512, 768, 4096), (300, 264, 3088)])
def test_fp8_gemm_vs_f32_product_of_the_same_fp8_operands(dev, a_fmt, M, N, K):
    from prompt_tts_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    a8 = torch.empty(M, K, dtype=torch.uint8, device=dev); w8 = torch.empty(N, K, dtype=torch.uint8, device=dev)
    sa = torch.empty(258, device=dev); sw = torch.empty(258, device=dev)
    ops.fp8_quantize(a, a8, sa, a_fmt); ops.fp8_quantize(w, w8, sw, 0)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8(M, N, K, a8, w8, out, sa, sw, a_format=a_fmt, bias=bias, residual=res, ldr=N)
    qa = a8.view(F8[a_fmt][0]).float(); qw = w8.view(F8[0][0]).float()
    want = (qa.double() @ qw.double().t()) * float(sa[1]) * float(sw[1]) + bias.double() + res.double()
    _assert_bf16_close(out, want)
    exact = a.double() @ w.double().t() + bias.double() + res.double()                  # what bf16 operands would give
    rel = float((out.double() - exact).pow(2).mean().sqrt() / (a.double() @ w.double().t()).pow(2).mean().sqrt())
    assert rel < (0.05 if a_fmt == 0 else 0.09), rel                                     # e4m3: 2^-4 / sqrt(3) per operand; e5m2 doubles it


def test_fp8_gemm_with_transposed_weight_copy_is_the_dgrad(dev):
    """dx = dy W through the transposed fp8 copy: A = dy (e5m2) [M][N], B = W^T [K][N]."""
    from prompt_tts_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 768, 1024                         # dy [M][N], W [N][K] -> dx [M][K]
    dy = (torch.randn(M, N, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    w8 = torch.empty(N, K, dtype=torch.uint8, device=dev); w8t = torch.empty(K, N, dtype=torch.uint8, device=dev)
    dy8 = torch.empty(M, N, dtype=torch.uint8, device=dev)
    sw = torch.empty(258, device=dev); sd = torch.empty(258, device=dev)
    ops.fp8_quantize(w, w8, sw, 0, out_t=w8t); ops.fp8_quantize(dy, dy8, sd, 1)
    dx = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8(M, K, N, dy8, w8t, dx, sd, sw, a_format=1)
    want = (dy8.view(torch.float8_e5m2).double() @ w8.view(torch.float8_e4m3fn).double()) * float(sd[1]) * float(sw[1])
    _assert_bf16_close(dx, want)


def test_fp8_entry_points_refuse_bad_arguments(dev):
    from prompt_tts_amd import ops
    x = torch.zeros(64, 24, dtype=torch.bfloat16, device=dev)           # cols % 16 != 0
    with pytest.raises(RuntimeError):
        ops.fp8_quantize(x, torch.empty(64, 24, dtype=torch.uint8, device=dev), torch.empty(258, device=dev))
    a8 = torch.zeros(256, 128, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError):                                    # K % 16 != 0
        ops.gemm_fp8(256, 256, 120, a8, a8, torch.empty(256, 256, dtype=torch.bfloat16, device=dev), torch.ones(258, device=dev), torch.ones(258, device=dev))


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), float("-inf")])
def test_fp8_quantiser_propagates_non_finite_values(dev, bad):
    """one NaN / Inf element used to come back as a finite clamp value (-448 / 0), so a diverged activation re-entered
    the fp8 GEMMs as finite garbage.  Now the published scale is NaN and the GEMM output is poisoned, as on the bf16 path."""
    from prompt_tts_amd import ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 256, 256, 128
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    a[17, 5] = bad
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    a8 = torch.empty(M, K, dtype=torch.uint8, device=dev); w8 = torch.empty(N, K, dtype=torch.uint8, device=dev)
    sa = torch.empty(258, device=dev); sw = torch.empty(258, device=dev)
    ops.fp8_quantize(a, a8, sa, 0); ops.fp8_quantize(w, w8, sw, 0)
    assert torch.isnan(sa[1]) and torch.isfinite(sw[1])
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8(M, N, K, a8, w8, out, sa, sw)
    assert torch.isnan(out.float()).all()
