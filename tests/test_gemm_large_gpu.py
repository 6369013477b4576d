"""Parity of the GEMM kernels at the sizes the benchmark actually runs (BASELINE configs[1]).

pick_tile (csrc/gemm.hip) moves bf16 forward / dgrad / conv GEMMs to the 256x256 eight-phase kernel once a problem has
>= 192 tiles of 256x256, and the weight-gradient GEMMs run split-K over a 32 768-row reduction: none of that is reached by the
small shapes of tests/test_kernels_gpu.py.  Every class the training step launches is checked here at such sizes, including
M / N / K that are not multiples of the 256 / 64 tile edges.

Reference: a plain PyTorch f32 matmul of the SAME bf16-rounded inputs, evaluated on the device (rocBLAS: an independent
implementation; a CPU matmul of these sizes would take minutes).  Conv references are built from explicit shifted copies of
the input (no im2col inside the kernel under test).  Tolerance: TOL below, as in test_kernels_gpu.py.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 1.5e-2}


def _ops():
    from prompt_tts_amd import ops, _lib
    return ops, _lib


def relerr(got, ref):
    got = got.detach().float(); ref = ref.detach().float()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


def rnd(shape, dtype, dev, gen, scale=1.0):
    x = (torch.randn(shape, generator=gen, device=dev) * scale).to(dtype)
    return x, x.float()


def _gen(dev, seed):
    return torch.Generator(device=dev).manual_seed(seed)


def tiles256(M, N):
    return ((M + 255) // 256) * ((N + 255) // 256)


# ---- forward NN / dgrad NT, plain operands --------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(32768, 512, 512), (32768, 1536, 512), (32768, 4096, 512), (32768, 512, 2048),
                                   (8200, 1544, 520), (12288, 4096, 2048)])
def test_forward_plain_large(dev, M, N, K):
    ops, L = _ops()
    dtype = torch.bfloat16
    assert tiles256(M, N) >= 192
    g = _gen(dev, 1)
    a, af = rnd((M, K), dtype, dev, g); w, wf = rnd((N, K), dtype, dev, g, K ** -0.5)
    r, rf = rnd((M, N), dtype, dev, g); bias = torch.randn(N, generator=g, device=dev)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
    ops.gemm(M, N, K, ops.plain(a), ops.plain(w), out, ops.pt_dtype(a), bias=bias, residual=r, ldr=N)
    assert relerr(out, af @ wf.t() + bias + rf) < TOL[dtype]


@pytest.mark.parametrize("M,Nout,Kin", [(32768, 512, 512), (32768, 1536, 512), (32768, 4096, 512), (32768, 512, 2048),
                                        (8200, 520, 1544)])
def test_dgrad_plain_large(dev, M, Nout, Kin):
    ops, L = _ops()
    dtype = torch.bfloat16
    assert tiles256(M, Kin) >= 192 or (M, Kin) == (32768, 512)
    g = _gen(dev, 2)
    dy, dyf = rnd((M, Nout), dtype, dev, g); w, wf = rnd((Nout, Kin), dtype, dev, g, Nout ** -0.5)
    acc, accf = rnd((M, Kin), dtype, dev, g)
    dx = acc.clone()
    ops.gemm(M, Kin, Nout, ops.plain(dy), ops.plain(w, trans=True), dx, ops._DT[dtype], residual2=dx, ldr2=Kin)  # in-place accumulate
    assert relerr(dx, dyf @ wf + accf) < TOL[dtype]


# ---- conv k3 forward (stride 1 / stride 2 / upsample-fused) and dgrad -----------------------------------------------------
def _conv_ref(xf, w3f, B, n_in, mode):
    """xf (B*n_in, cin) f32 token-major; w3f [cout][3][cin] f32.  Returns (B*n_out, cout) f32."""
    cin = xf.shape[1]
    x = xf.view(B, n_in, cin)
    if mode == "up2":
        x = x.repeat_interleave(2, dim=1)
    z = torch.zeros(B, 1, cin, device=xf.device)
    xp = torch.cat([z, x, z], dim=1)                                      # pad 1 on both sides
    n = x.shape[1]
    cols = torch.cat([xp[:, 0:n], xp[:, 1:n + 1], xp[:, 2:n + 2]], dim=2)   # tap-major [.., 3*cin]
    if mode == "s2":
        cols = cols[:, ::2]
    return cols.reshape(-1, 3 * cin) @ w3f.reshape(w3f.shape[0], -1).t()


@pytest.mark.parametrize("mode,B,n_in,cin,cout", [("s1", 32, 1024, 512, 512), ("s2", 64, 1024, 512, 512),
                                                  ("up2", 32, 512, 512, 512), ("s1", 25, 1000, 520, 264),
                                                  ("s1", 32, 1024, 1024, 512)])
def test_conv_forward_large(dev, mode, B, n_in, cin, cout):
    ops, L = _ops()
    dtype = torch.bfloat16
    g = _gen(dev, 3)
    n_out = {"s1": n_in, "s2": (n_in - 1) // 2 + 1, "up2": 2 * n_in}[mode]
    rowmap = {"s1": L.PT_MAP_S1, "s2": L.PT_MAP_S2, "up2": L.PT_MAP_UP2}[mode]
    assert tiles256(B * n_out, cout) >= 192
    x, xf = rnd((B * n_in, cin), dtype, dev, g); w3, w3f = rnd((cout, 3, cin), dtype, dev, g, (3 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g, device=dev); temb = torch.randn(B, cout, generator=g, device=dev)
    res, resf = rnd((B * n_out, cout), dtype, dev, g)
    out = torch.full((B * n_out, cout), float("nan"), dtype=dtype, device=dev)
    ops.gemm(B * n_out, cout, 3 * cin, ops.conv(x, cin, n_out, n_in, rowmap), ops.plain(w3.view(cout, 3 * cin)), out, ops._DT[dtype],
             bias=bias, row_bias=temb, row_bias_rows=n_out, residual=res, ldr=cout)
    ref = _conv_ref(xf, w3f, B, n_in, mode) + bias + temb.repeat_interleave(n_out, dim=0) + resf
    assert relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("mode,B,n_in,cin,cout", [("s1", 32, 1024, 512, 512), ("s2", 64, 1024, 512, 512),
                                                  ("s1", 25, 1000, 264, 520)])
def test_conv_dgrad_large(dev, mode, B, n_in, cin, cout):
    """dx = conv^T(dy): checked through autograd of the explicit-copies reference."""
    ops, L = _ops()
    dtype = torch.bfloat16
    g = _gen(dev, 4)
    n_out = n_in if mode == "s1" else (n_in - 1) // 2 + 1
    assert tiles256(B * n_in, cin) >= 192
    w3, w3f = rnd((cout, 3, cin), dtype, dev, g, (3 * cin) ** -0.5)
    dy, dyf = rnd((B * n_out, cout), dtype, dev, g)
    xz = torch.zeros(B * n_in, cin, device=dev, requires_grad=True)
    _conv_ref(xz, w3f, B, n_in, mode).backward(dyf)
    dx = torch.full((B * n_in, cin), float("nan"), dtype=dtype, device=dev)
    rowmap = L.PT_MAP_S1 if mode == "s1" else L.PT_MAP_S2_DGRAD
    ops.gemm(B * n_in, cin, 3 * cout, ops.conv(dy, cout, n_in, n_out, rowmap), ops.wflip(w3, cout, cin), dx, ops._DT[dtype])
    assert relerr(dx, xz.grad) < TOL[dtype]


# ---- weight gradients at the benchmark's reduction length (split-K) ---------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("Mred,Nout,Kin", [(32768, 512, 512), (32768, 1536, 512), (32768, 4096, 512), (32768, 512, 2048),
                                           (16384, 512, 1024), (8200, 520, 264)])
def test_wgrad_plain_large(dev, dtype, Mred, Nout, Kin):
    from prompt_tts_amd import engine as E
    ops, L = _ops()
    if dtype == torch.float32 and Nout * Kin > 1536 * 512:
        pytest.skip("f32 parity mode is exercised at the smaller shapes")
    g = _gen(dev, 5)
    dy, dyf = rnd((Mred, Nout), dtype, dev, g); x, xf = rnd((Mred, Kin), dtype, dev, g)
    dw = torch.ones(Nout, Kin, dtype=torch.float32, device=dev)            # accumulates on top of existing gradients
    gb = torch.zeros(Nout, dtype=torch.float32, device=dev)
    ops.gemm(Nout, Kin, Mred, ops.plain(dy, trans=True), ops.plain(x, trans=True), dw, ops._DT[dtype],
             out_kind=L.PT_OUT_F32_ATOMIC, split_k=E._split_k(Nout, Kin, Mred, dtype), arow_sum=gb, arow_n=Nout)
    ref = dyf.t() @ xf
    assert relerr(dw - 1.0, ref) < TOL[dtype]
    assert relerr(gb, dyf.sum(0)) < TOL[dtype]


@pytest.mark.parametrize("mode,B,n_in,cin,cout", [("s1", 32, 1024, 512, 512), ("s1", 32, 1024, 1024, 512),
                                                  ("s2", 32, 1024, 512, 512), ("up2", 32, 512, 512, 512)])
def test_wgrad_conv_large(dev, mode, B, n_in, cin, cout):
    from prompt_tts_amd import engine as E
    ops, L = _ops()
    dtype = torch.bfloat16
    g = _gen(dev, 6)
    n_out = {"s1": n_in, "s2": (n_in - 1) // 2 + 1, "up2": 2 * n_in}[mode]
    rowmap = {"s1": L.PT_MAP_S1, "s2": L.PT_MAP_S2, "up2": L.PT_MAP_UP2}[mode]
    x, xf = rnd((B * n_in, cin), dtype, dev, g); dy, dyf = rnd((B * n_out, cout), dtype, dev, g)
    wz = torch.zeros(cout, 3, cin, device=dev, requires_grad=True)
    _conv_ref(xf, wz, B, n_in, mode).backward(dyf)
    dw = torch.zeros(cout, 3 * cin, dtype=torch.float32, device=dev)
    gb = torch.zeros(cout, dtype=torch.float32, device=dev)
    ops.gemm(cout, 3 * cin, B * n_out, ops.plain(dy, trans=True), ops.conv(x, cin, n_out, n_in, rowmap, trans=True), dw,
             ops._DT[dtype], ldc=3 * cin, out_kind=L.PT_OUT_F32_ATOMIC, split_k=E._split_k(cout, 3 * cin, B * n_out, dtype),
             arow_sum=gb, arow_n=cout)
    assert relerr(dw.view(cout, 3, cin), wz.grad) < TOL[dtype]
    assert relerr(gb, dyf.sum(0)) < TOL[dtype]


# ---- grouped weight gradients (pt_wgrad_group): one launch for the weight gradients of a transformer block / resnet ---------
def _ws(dev, target=256):
    ops, L = _ops()
    return torch.empty(ops.wgrad_group_ws_floats(target), dtype=torch.float32, device=dev)


@pytest.mark.parametrize("target", [256, 64])
def test_wgrad_group_transformer_block(dev, target):
    """qkv, out, q (cross), kv (cross, shorter reduction), ff1, ff2 of one BasicTransformerBlock at config B size + one ragged
    problem, as ONE grouped launch; gradients accumulate on top of existing values; bias gradients ride along."""
    ops, L = _ops()
    dtype = torch.bfloat16
    g = _gen(dev, 7)
    T, S = 32768, 8192
    probs = [(T, 1536, 512, False), (T, 512, 512, True), (T, 512, 512, False), (S, 1024, 512, False), (T, 4096, 512, True),
             (T, 512, 2048, True), (8200, 520, 264, True)]
    keep, descs, checks = [], [], []
    for red, nout, kin, has_bias in probs:
        dy, dyf = rnd((red, nout), dtype, dev, g); x, xf = rnd((red, kin), dtype, dev, g)
        dw = torch.full((nout, kin), 0.5, dtype=torch.float32, device=dev)
        gb = torch.zeros(nout, dtype=torch.float32, device=dev) if has_bias else None
        descs.append(ops.gemm_desc(nout, kin, red, ops.plain(dy, trans=True), ops.plain(x, trans=True), dw,
                                   out_kind=L.PT_OUT_F32_ATOMIC, arow_sum=gb, arow_n=nout if has_bias else 0))
        keep.append((dy, x)); checks.append((dw, gb, dyf, xf))
    ws = _ws(dev, target)
    if target < 256:
        with pytest.raises(RuntimeError):                    # more 256 x 256 tiles than workgroups: refused, nothing launched
            ops.wgrad_group(descs, ws, target)
        return
    ops.wgrad_group(descs, ws, target)
    for dw, gb, dyf, xf in checks:
        assert relerr(dw - 0.5, dyf.t() @ xf) < TOL[dtype]
        if gb is not None:
            assert relerr(gb, dyf.sum(0)) < TOL[dtype]
    # single-problem groups give the same numbers as the grouped launch (different split counts)
    dy, x = keep[1]
    dw1 = torch.zeros(512, 512, dtype=torch.float32, device=dev)
    ops.wgrad_group([ops.gemm_desc(512, 512, T, ops.plain(dy, trans=True), ops.plain(x, trans=True), dw1,
                                   out_kind=L.PT_OUT_F32_ATOMIC)], ws)
    assert relerr(dw1, checks[1][0] - 0.5) < 1e-5


def test_wgrad_group_resnet_convs(dev):
    """conv1 (1024 -> 512, the up-path concat width), conv2, a stride-2 downsampler and an upsample-fused conv: the conv-gather
    operand class of the grouped kernel (explicit-copies reference through autograd)."""
    ops, L = _ops()
    dtype = torch.bfloat16
    g = _gen(dev, 8)
    B = 32
    cases = [("s1", 1024, 1024, 512), ("s1", 1024, 512, 512), ("s2", 1024, 512, 512), ("up2", 512, 512, 512), ("s1", 1000, 264, 520)]
    descs, checks, keep = [], [], []
    for mode, n_in, cin, cout in cases:
        Bc = B if n_in != 1000 else 5
        n_out = {"s1": n_in, "s2": (n_in - 1) // 2 + 1, "up2": 2 * n_in}[mode]
        rowmap = {"s1": L.PT_MAP_S1, "s2": L.PT_MAP_S2, "up2": L.PT_MAP_UP2}[mode]
        x, xf = rnd((Bc * n_in, cin), dtype, dev, g); dy, dyf = rnd((Bc * n_out, cout), dtype, dev, g)
        wz = torch.zeros(cout, 3, cin, device=dev, requires_grad=True)
        _conv_ref(xf, wz, Bc, n_in, mode).backward(dyf)
        dw = torch.zeros(cout, 3 * cin, dtype=torch.float32, device=dev)
        gb = torch.zeros(cout, dtype=torch.float32, device=dev)
        descs.append(ops.gemm_desc(cout, 3 * cin, Bc * n_out, ops.plain(dy, trans=True),
                                   ops.conv(x, cin, n_out, n_in, rowmap, trans=True), dw, ldc=3 * cin,
                                   out_kind=L.PT_OUT_F32_ATOMIC, arow_sum=gb, arow_n=cout))
        keep.append((x, dy)); checks.append((dw, gb, wz.grad, dyf, cout, cin))
    ops.wgrad_group(descs, _ws(dev))
    for dw, gb, ref, dyf, cout, cin in checks:
        assert relerr(dw.view(cout, 3, cin), ref) < TOL[dtype]
        assert relerr(gb, dyf.sum(0)) < TOL[dtype]
    # one group holds ONE operand class: a plain problem next to conv problems is refused
    mixed = [descs[0], ops.gemm_desc(512, 512, 1024, ops.plain(keep[1][1], trans=True), ops.plain(keep[1][0], trans=True),
                                     checks[1][0], out_kind=L.PT_OUT_F32_ATOMIC)]
    with pytest.raises(RuntimeError):
        ops.wgrad_group(mixed, _ws(dev))


@pytest.mark.parametrize("B,n,cin,cout", [(32, 1024, 512, 512), (6, 320, 1024, 512), (2, 192, 256, 384), (1, 512, 512, 512)])
def test_conv_wgrad_flat_item_with_boundary_corrections(dev, B, n, cin, cout, monkeypatch):
    """engine.conv3_bwd's stride-1 weight gradient walks ONE flat item of B n rows (constant address step) and subtracts the
    two cross-item products per conv that the zero padding excludes (rank B-1 problems in the plain group): against the
    autograd gradient of the explicit-copies conv, and against the per-item row map (PT_CONV_WGRAD_FLAT=0)."""
    ops, L = _ops()
    from prompt_tts_amd import engine as E
    dtype = torch.bfloat16
    g = _gen(dev, 21 + B)
    x, xf = rnd((B * n, cin), dtype, dev, g); dy, dyf = rnd((B * n, cout), dtype, dev, g)
    w3 = torch.zeros(cout, 3 * cin, dtype=dtype, device=dev)
    wz = torch.zeros(cout, 3, cin, device=dev, requires_grad=True)
    _conv_ref(xf, wz, B, n, "s1").backward(dyf)
    got = {}
    for flat in (True, False):
        monkeypatch.setattr(E, "CONV_WGRAD_FLAT", flat)
        gw = torch.zeros(cout, 3 * cin, dtype=torch.float32, device=dev); gb = torch.zeros(cout, dtype=torch.float32, device=dev)
        E.conv3_bwd(dy, x, w3, gw, gb, B, n, n, need_dx=False)
        E.flush_wgrads(); E.join_side_stream(dev); torch.cuda.synchronize()
        assert relerr(gw.view(cout, 3, cin), wz.grad) < TOL[dtype]
        assert relerr(gb, dyf.sum(0)) < TOL[dtype]
        got[flat] = gw
    # both forms sum the same products (other orders): far inside the bf16 tolerance of either against the reference
    assert relerr(got[True], got[False]) < 1e-4


# ---- GEGLU fused into the ff1 GEMM epilogue (act 2) and into the ff2 dgrad epilogue (act 3) ----------------------------------
def _interleave_rows(w, F):
    """[2F][...] value rows | gate rows -> the interleaved order of pt_gemm act 2 (row 64q+t: t<32 value 32q+t, else gate)."""
    idx = torch.arange(2 * F, device=w.device)
    q, t = idx // 64, idx % 64
    src = torch.where(t < 32, 32 * q + t, F + 32 * q + (t - 32))
    return w[src].contiguous(), src


@pytest.mark.parametrize("M,d,F", [(32768, 512, 2048), (512, 256, 1024), (8192, 512, 256)])
def test_geglu_fused_forward_backward(dev, M, d, F):
    """FeedForward of the reference's BasicTransformerBlock (diffusers GEGLU: Linear(d, 2F) -> value * gelu_erf(gate)) through
    the fused epilogues, against plain PyTorch f32 on the same bf16-rounded inputs; the 512-row case runs on the 128 x 128
    kernel, the others on the eight-phase kernel."""
    import torch.nn.functional as Fn
    from prompt_tts_amd import engine as E
    ops, L = _ops()
    dtype = torch.bfloat16
    g = _gen(dev, 9)
    x, xf = rnd((M, d), dtype, dev, g); w1, w1f = rnd((2 * F, d), dtype, dev, g, d ** -0.5)
    b1 = torch.randn(2 * F, generator=g, device=dev) * 0.1
    w1i, src = _interleave_rows(w1, F)
    proj = torch.full((M, 2 * F), float("nan"), dtype=dtype, device=dev); act = torch.full((M, F), float("nan"), dtype=dtype, device=dev)
    ops.gemm(M, 2 * F, d, ops.plain(x), ops.plain(w1i), proj, ops._DT[dtype], bias=b1, act=2, out2=act, ldc2=F)
    pref = xf @ w1f.t() + b1
    assert relerr(proj, pref[:, src]) < TOL[dtype]                       # raw projection, interleaved columns, bias in place
    aref = pref[:, :F] * Fn.gelu(pref[:, F:])
    assert relerr(act, aref) < TOL[dtype]
    # backward of the following Linear(F, d): d(act) = dout W2 never leaves the kernel; d(proj) comes out interleaved
    dout, doutf = rnd((M, d), dtype, dev, g); w2, w2f = rnd((d, F), dtype, dev, g, F ** -0.5)
    dproj = torch.full((M, 2 * F), float("nan"), dtype=dtype, device=dev)
    ops.gemm(M, F, d, ops.plain(dout), ops.plain(w2, trans=True), dproj, ops._DT[dtype], ldc=2 * F, act=3, residual=proj, ldr=2 * F)
    pr = proj.float()[:, torch.argsort(src)].clone().requires_grad_(True)          # the stored (bf16-rounded) projection, un-interleaved
    (pr[:, :F] * Fn.gelu(pr[:, F:])).backward(doutf @ w2f)
    assert relerr(dproj, pr.grad[:, src]) < TOL[dtype]
    # the stand-alone interleaved kernels (row counts the fused epilogue does not take) agree with the fused ones
    proj2 = torch.empty_like(proj); act2 = torch.empty_like(act)
    ops.gemm(M, 2 * F, d, ops.plain(x), ops.plain(w1i), proj2, ops._DT[dtype])
    ops.geglu_fwd(proj2, act2, bias=b1, interleaved=True)
    assert relerr(proj2, proj.float()) < 1e-2 and relerr(act2, act.float()) < 1e-2
    dact = torch.empty(M, F, dtype=dtype, device=dev)
    ops.gemm(M, F, d, ops.plain(dout), ops.plain(w2, trans=True), dact, ops._DT[dtype])
    dproj2 = torch.empty_like(dproj); ops.geglu_bwd(dact, proj, dproj2, interleaved=True)
    assert relerr(dproj2, dproj.float()) < 2e-2
    # weight / bias gradient of the interleaved projection through the grouped wgrad: written in the ORIGINAL row order
    gw = torch.zeros(2 * F, d, dtype=torch.float32, device=dev); gb = torch.zeros(2 * F, dtype=torch.float32, device=dev)
    ws = torch.empty(ops.wgrad_group_ws_floats(256), dtype=torch.float32, device=dev)
    ops.wgrad_group([ops.gemm_desc(2 * F, d, M, ops.plain(dproj, trans=True), ops.plain(x, trans=True), gw, out_kind=L.PT_OUT_F32_ATOMIC,
                                   arow_sum=gb, arow_n=2 * F, geglu_rows=F)], ws)
    dpo = dproj.float()[:, torch.argsort(src)]
    assert relerr(gw, dpo.t() @ xf) < TOL[dtype] and relerr(gb, dpo.sum(0)) < TOL[dtype]
    # misuse is refused before any launch
    with pytest.raises(RuntimeError):
        ops.gemm(M + 8, 2 * F, d, ops.plain(x), ops.plain(w1i), proj, ops._DT[dtype], bias=b1, act=2, out2=act, ldc2=F)
