"""Kernel-level parity: every C-ABI entry point against plain PyTorch fp32/fp64 on the CPU.

Tolerances (written here, used below):
  f32 path : max|err| <= 1e-4 * max|ref|   (north_star asks 1e-3 relative; exact-f32 MFMA does better)
  bf16 path: inputs are rounded to bf16 first and the reference is computed in f32 on those rounded inputs;
             max|err| <= 1.5e-2 * max|ref| (bf16 has 8 significant bits: output rounding alone is 2^-9)
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 1.5e-2}
DTYPES = [torch.float32, torch.bfloat16]


def _ops():
    from prompt_tts_amd import ops, _lib
    return ops, _lib


def rnd(shape, dtype, dev, gen, scale=1.0):
    x = (torch.randn(shape, generator=gen) * scale).to(dtype)
    return x.to(dev), x.float()


def relerr(got, ref):
    got = got.detach().float().cpu(); ref = ref.detach().float().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 136), (1024, 512, 512), (77, 8, 24)])
def test_gemm_nt_bias_residual(dev, dtype, M, N, K):
    ops, L = _ops()
    g = torch.Generator().manual_seed(1)
    a, af = rnd((M, K), dtype, dev, g); w, wf = rnd((N, K), dtype, dev, g, K ** -0.5)
    r, rf = rnd((M, N), dtype, dev, g); bias = torch.randn(N, generator=g)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
    ops.gemm(M, N, K, ops.plain(a), ops.plain(w), out, ops.pt_dtype(a), bias=bias.to(dev), residual=r, ldr=N)
    ref = af @ wf.t() + bias + rf
    assert relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_exact_integers_asymmetric(dev, dtype):
    """A = I-like / small integers: catches transposed or permuted MFMA layouts exactly."""
    ops, L = _ops()
    M, N, K = 256, 256, 128
    a = torch.zeros(M, K); a[torch.arange(M), torch.arange(M) % K] = 1.0; a[:, 3] += 2.0
    w = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5) % 7 - 3.0   # asymmetric
    out = torch.empty(M, N, dtype=dtype, device=dev)
    ad, wd = a.to(dtype).to(dev), w.to(dtype).to(dev)       # keep the device buffers alive across the launch
    ops.gemm(M, N, K, ops.plain(ad), ops.plain(wd), out, ops._DT[dtype])
    assert torch.equal(out.float().cpu(), a @ w.t())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_dgrad_nn(dev, dtype):
    ops, L = _ops()
    g = torch.Generator().manual_seed(2)
    M, Nout, Kin = 260, 192, 320
    dy, dyf = rnd((M, Nout), dtype, dev, g); w, wf = rnd((Nout, Kin), dtype, dev, g, Nout ** -0.5)
    dx = torch.empty(M, Kin, dtype=dtype, device=dev)
    ops.gemm(M, Kin, Nout, ops.plain(dy), ops.plain(w, trans=True), dx, ops._DT[dtype])
    assert relerr(dx, dyf @ wf) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("split", [1, 4])
def test_gemm_wgrad_tt_atomic(dev, dtype, split):
    ops, L = _ops()
    g = torch.Generator().manual_seed(3)
    Mred, Nout, Kin = 1000, 136, 200
    dy, dyf = rnd((Mred, Nout), dtype, dev, g); x, xf = rnd((Mred, Kin), dtype, dev, g)
    dw = torch.ones(Nout, Kin, dtype=torch.float32, device=dev)          # accumulate on top of existing grads
    ops.gemm(Nout, Kin, Mred, ops.plain(dy, trans=True), ops.plain(x, trans=True), dw, ops._DT[dtype],
             out_kind=L.PT_OUT_F32_ATOMIC, split_k=split)
    ref = dyf.t() @ xf + 1.0
    assert relerr(dw, ref) < TOL[dtype]
    # the bias gradient rides on the same GEMM (all-ones MFMA column): replicated destination, first 130 of 136 rows only
    dw2 = torch.zeros(Nout, Kin, dtype=torch.float32, device=dev)
    rep = torch.full((3, 192), 0.5, device=dev)
    ops.gemm(Nout, Kin, Mred, ops.plain(dy, trans=True), ops.plain(x, trans=True), dw2, ops._DT[dtype],
             out_kind=L.PT_OUT_F32_ATOMIC, split_k=split, arow_sum=rep[0], arow_n=130, arow_rep=3, arow_stride=192)
    assert relerr(dw2, ref - 1.0) < TOL[dtype]
    got = rep.sum(0) - 1.5
    assert relerr(got[:130], dyf.sum(0)[:130]) < TOL[dtype] and float(got[130:].abs().max()) == 0.0


def _tok(x):   # (B,C,N) -> token-major (B*N, C)
    return x.permute(0, 2, 1).reshape(-1, x.shape[1]).contiguous()


def _untok(y, B):
    return y.view(B, -1, y.shape[1]).permute(0, 2, 1)


def _wshadow(w, cin_pad=None):   # (Cout,Cin,3) -> [Cout][3][cin_pad]
    co, ci, _ = w.shape
    cp = cin_pad or ci
    s = torch.zeros(co, 3, cp, dtype=w.dtype)
    s[:, :, :ci] = w.permute(0, 2, 1)
    return s.contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["s1", "s2", "up2"])
def test_conv_fwd(dev, dtype, mode):
    ops, L = _ops()
    g = torch.Generator().manual_seed(4)
    B, Cin, Cout, Nin = 3, 64, 96, 50
    x = torch.randn(B, Cin, Nin, generator=g).to(dtype); w = (torch.randn(Cout, Cin, 3, generator=g) * (3 * Cin) ** -0.5).to(dtype)
    bias = torch.randn(Cout, generator=g); temb = torch.randn(B, Cout, generator=g)
    xf, wf = x.float(), w.float()
    if mode == "s1":
        ref = F.conv1d(xf, wf, bias, padding=1); rowmap = L.PT_MAP_S1
    elif mode == "s2":
        ref = F.conv1d(xf, wf, bias, stride=2, padding=1); rowmap = L.PT_MAP_S2
    else:
        ref = F.conv1d(F.interpolate(xf, scale_factor=2.0, mode="nearest"), wf, bias, padding=1); rowmap = L.PT_MAP_UP2
    ref = ref + temb[:, :, None]
    Nout = ref.shape[2]
    xt = _tok(x).to(dev); ws = _wshadow(w).view(Cout, 3 * Cin).to(dev)
    out = torch.empty(B * Nout, Cout, dtype=dtype, device=dev)
    ops.gemm(B * Nout, Cout, 3 * Cin, ops.conv(xt, Cin, Nout, Nin, rowmap), ops.plain(ws), out, ops._DT[dtype],
             bias=bias.to(dev), row_bias=temb.to(dev), row_bias_rows=Nout)
    assert relerr(_untok(out, B), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["s1", "s2"])
def test_conv_dgrad_wgrad(dev, dtype, mode):
    ops, L = _ops()
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, Nin = 2, 64, 128, 48
    stride = 1 if mode == "s1" else 2
    x = torch.randn(B, Cin, Nin, generator=g).to(dtype); w = (torch.randn(Cout, Cin, 3, generator=g) * (3 * Cin) ** -0.5).to(dtype)
    xf = x.float().requires_grad_(True); wf = w.float().requires_grad_(True)
    y = F.conv1d(xf, wf, None, stride=stride, padding=1)
    Nout = y.shape[2]
    dy = torch.randn(B, Cout, Nout, generator=g).to(dtype)
    y.backward(dy.float())
    dyt = _tok(dy).to(dev); xt = _tok(x).to(dev); ws = _wshadow(w).to(dev)
    # dgrad: dx[(b,m), ci] = sum_{tap',co} dY[(b, src(m,tap')), co] * W[co][2-tap'][ci]
    dx = torch.empty(B * Nin, Cin, dtype=dtype, device=dev)
    rowmap = L.PT_MAP_S1 if stride == 1 else L.PT_MAP_S2_DGRAD
    ops.gemm(B * Nin, Cin, 3 * Cout, ops.conv(dyt, Cout, Nin, Nout, rowmap), ops.wflip(ws, Cout, Cin), dx, ops._DT[dtype])
    assert relerr(_untok(dx, B), xf.grad) < TOL[dtype]
    # wgrad in the kernel layout [Cout][3][Cin]
    dw = torch.zeros(Cout, 3 * Cin, dtype=torch.float32, device=dev)
    rm = L.PT_MAP_S1 if stride == 1 else L.PT_MAP_S2
    ops.gemm(Cout, 3 * Cin, B * Nout, ops.plain(dyt, trans=True), ops.conv(xt, Cin, Nout, Nin, rm, trans=True), dw,
             ops._DT[dtype], out_kind=L.PT_OUT_F32_ATOMIC, split_k=2)
    assert relerr(dw.view(Cout, 3, Cin).permute(0, 2, 1), wf.grad) < TOL[dtype]
    # padded-Cin variant (conv_in): x carries 8 channels of which 5 are real; the stored gradient is [Cout][3][5]
    x8 = torch.zeros(B, 8, Nin).to(dtype); x8[:, :5] = x[:, :5]
    w5 = wf.detach()[:, :5].clone().requires_grad_(True)
    F.conv1d(x8[:, :5].float(), w5, None, stride=stride, padding=1).backward(dy.float())
    dw5 = torch.zeros(Cout, 15, dtype=torch.float32, device=dev)
    x8t = _tok(x8).to(dev)
    ops.gemm(Cout, 24, B * Nout, ops.plain(dyt, trans=True), ops.conv(x8t, 8, Nout, Nin, rm, trans=True), dw5,
             ops._DT[dtype], out_kind=L.PT_OUT_F32_ATOMIC, conv_wgrad_cin=8, conv_wgrad_cin_store=5)
    assert relerr(dw5.view(Cout, 3, 5).permute(0, 2, 1), w5.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_concat_operand(dev, dtype):
    ops, L = _ops()
    g = torch.Generator().manual_seed(6)
    M, C1, C2, N = 200, 64, 128, 72
    a1, a1f = rnd((M, C1), dtype, dev, g); a2, a2f = rnd((M, C2), dtype, dev, g); w, wf = rnd((N, C1 + C2), dtype, dev, g, 0.1)
    out = torch.empty(M, N, dtype=dtype, device=dev)
    ops.gemm(M, N, C1 + C2, ops.concat(a1, a2), ops.plain(w), out, ops._DT[dtype])
    assert relerr(out, torch.cat([a1f, a2f], 1) @ wf.t()) < TOL[dtype]
    # wgrad against the concat input (shortcut conv 1x1): dW[N][C1+C2] = dY^T [X1|X2]
    dy, dyf = rnd((M, N), dtype, dev, g)
    dw = torch.zeros(N, C1 + C2, dtype=torch.float32, device=dev)
    ops.gemm(N, C1 + C2, M, ops.plain(dy, trans=True), ops.concat(a1, a2, trans=True), dw, ops._DT[dtype],
             out_kind=L.PT_OUT_F32_ATOMIC)
    assert relerr(dw, dyf.t() @ torch.cat([a1f, a2f], 1)) < TOL[dtype]


@pytest.mark.parametrize("M,N,K", [(300, 136, 200), (1024, 512, 896), (4096, 256, 64)])
def test_gemm_f32_x3_is_f32_class(dev, M, N, K):
    """pt_gemm_desc.f32_x3: f32 operands multiplied as a bf16 x 3 split (hi hi + hi lo + lo hi on the bf16 MFMA).  Against an f64
    product: error ~2^-16 per product -> well below 1e-4 of the result's scale, where a plain bf16 product sits at 4e-3; the
    exact-f32 form (x3 = 0, the training parity mode) stays at 1e-6.  Same for a conv-gather operand with the ELU epilogue."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * K ** -0.5; bias = torch.randn(N, generator=g)
    want = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    errs = {}
    for x3 in (False, True):
        out = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, ops.plain(ad), ops.plain(wd), out, L.PT_F32, bias=bd, x3=x3)
        errs[x3] = float((out.cpu().double() - want).abs().max() / want.abs().max())
    assert errs[False] < 2e-6 and errs[True] < 5e-5, errs
    if K % 8 == 0 and M % 64 == 0:
        Bb, n = 4, M // 4
        cin = K
        w3 = torch.randn(N, 3 * cin, generator=g) * (3 * cin) ** -0.5
        x = a.view(Bb, n, cin)
        xp = torch.cat([x[:, 2:3].flip(1), x[:, 1:2], x], 1)                       # causal reflect padding (k = 3): rows -2, -1 -> 2, 1
        cols = torch.cat([xp[:, t:t + n] for t in range(3)], -1).reshape(M, 3 * cin)
        ref = torch.nn.functional.elu(cols.double() @ w3.double().t() + bias.double())
        out = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, 3 * cin, ops.conv(ad, cin, n, n, L.PT_MAP_CAUSAL_REFLECT, taps=3), ops.plain(w3.to(dev)), out, L.PT_F32, bias=bd,
                 act=1, x3=True)
        assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < 5e-5


def _attn_ref(q, k, v, scale, causal=False, kv_len=None):
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    if kv_len is not None:
        idx = torch.arange(k.shape[2])[None, None, None, :]
        s = s.masked_fill(idx >= kv_len[:, None, None, None], float("-inf"))
    if causal:
        tri = torch.ones(q.shape[2], k.shape[2], dtype=torch.bool).tril()
        s = s.masked_fill(~tri, float("-inf"))
    return torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), v), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Nq,Nk,D,causal,masked", [
    (2, 2, 128, 128, 64, False, False), (2, 3, 200, 77, 64, False, False), (1, 2, 256, 256, 32, False, False),
    (1, 2, 130, 192, 128, False, False), (2, 2, 160, 160, 64, True, False), (3, 2, 96, 100, 64, False, True)])
def test_attention_fwd_bwd(dev, dtype, B, H, Nq, Nk, D, causal, masked):
    ops, L = _ops()
    g = torch.Generator().manual_seed(7)
    C = H * D
    q, qf = rnd((B * Nq, C), dtype, dev, g); k, kf = rnd((B * Nk, C), dtype, dev, g); v, vf = rnd((B * Nk, C), dtype, dev, g)
    do, dof = rnd((B * Nq, C), dtype, dev, g)
    kv_len = torch.tensor([Nk, 37, 64][:B], dtype=torch.int32) if masked else None
    scale = D ** -0.5
    def heads(x, n):
        return x.view(B, n, H, D).permute(0, 2, 1, 3).clone().requires_grad_(True)
    qh, kh, vh = heads(qf, Nq), heads(kf, Nk), heads(vf, Nk)
    oref, lseref = _attn_ref(qh, kh, vh, scale, causal, kv_len.long() if masked else None)
    oref.backward(heads(dof, Nq).detach())
    def flat(x, n):
        return x.permute(0, 2, 1, 3).reshape(B * n, C)
    o = torch.empty_like(q); lse = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
    kvd = kv_len.to(dev) if masked else None
    ops.attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, D, scale, causal, kvd)
    assert relerr(o, flat(oref, Nq)) < TOL[dtype]
    assert relerr(lse, lseref) < max(TOL[dtype], 2e-3 if dtype == torch.bfloat16 else 1e-4)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    delta = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
    ops.attn_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, Nq, Nk, D, scale, causal, kvd)
    tol = TOL[dtype] * (1 if dtype == torch.float32 else 2)
    assert relerr(dq, flat(qh.grad, Nq)) < tol
    assert relerr(dk, flat(kh.grad, Nk)) < tol
    assert relerr(dv, flat(vh.grad, Nk)) < tol


def test_attention_at_benchmark_size_items_independent(dev):
    """B H = 256, N = 1024, D = 64 (configs[1]'s self-attention, bf16).  Every (batch, head) is independent: items computed inside
    the big batch equal the same items computed alone BITWISE (same kernels, same tile schedule), and one item matches the f32
    reference.  Full-size buffers also put the operands at addresses small tensors never reach (the LDS-DMA path builds 64-bit
    addresses from two 32-bit halves: a sign extension there faulted only for tensors above a 2 GiB boundary)."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(21)
    B, H, N, D = 32, 8, 1024, 64
    C = H * D
    pad = torch.empty(3 << 30, dtype=torch.uint8, device=dev)          # push the operands past a 2 GiB boundary
    mk = lambda: (torch.randn(B * N, C, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    q, k, v, do = mk(), mk(), mk(), mk()
    del pad
    scale = D ** -0.5
    def run(qq, kk, vv, dd, b):
        o = torch.empty_like(qq); lse = torch.empty(b, H, N, dtype=torch.float32, device=dev); delta = torch.empty_like(lse)
        dq, dk, dv = torch.empty_like(qq), torch.empty_like(kk), torch.empty_like(vv)
        ops.attn_fwd(qq, kk, vv, o, lse, b, H, N, N, D, scale)
        ops.attn_bwd(qq, kk, vv, o, lse, dd, delta, dq, dk, dv, b, H, N, N, D, scale)
        return o, lse, dq, dk, dv
    big = run(q, k, v, do, B)
    for b in (0, 31):
        sl = slice(b * N, (b + 1) * N)
        one = run(q[sl].clone(), k[sl].clone(), v[sl].clone(), do[sl].clone(), 1)
        for name, x, y in zip(("o", "lse", "dq", "dk", "dv"), big, one):
            xx = x[b:b + 1] if name == "lse" else x[sl]
            assert torch.equal(xx, y), (b, name)
    def heads(x):
        return x[:N].float().cpu().view(1, N, H, D).permute(0, 2, 1, 3).clone().requires_grad_(True)
    qh, kh, vh = heads(q), heads(k), heads(v)
    oref, lseref = _attn_ref(qh, kh, vh, scale)
    oref.backward(heads(do).detach())
    flat = lambda x: x.permute(0, 2, 1, 3).reshape(N, C)
    assert relerr(big[0][:N], flat(oref)) < TOL[torch.bfloat16] and relerr(big[1][:1], lseref) < 2e-3
    assert relerr(big[2][:N], flat(qh.grad)) < 2 * TOL[torch.bfloat16]
    assert relerr(big[3][:N], flat(kh.grad)) < 2 * TOL[torch.bfloat16] and relerr(big[4][:N], flat(vh.grad)) < 2 * TOL[torch.bfloat16]


def test_attention_column_slices_and_rescale_branch(dev):
    """q/k/v as column slices of one fused [M,3C] buffer; one key row spiked so the running max jumps mid-stream."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(8)
    B, H, N, D = 1, 2, 192, 64
    C = H * D
    qkv = torch.randn(B * N, 3 * C, generator=g)
    qkv[150, C:2 * C] *= 12.0     # key 150 (third 64-key tile) dominates: forces a late rescale of O
    qd = qkv.to(dev)
    q, k, v = qd[:, :C], qd[:, C:2 * C], qd[:, 2 * C:]
    o = torch.empty(B * N, C, device=dev); lse = torch.empty(B, H, N, device=dev)
    ops.attn_fwd(q, k, v, o, lse, B, H, N, N, D, D ** -0.5)
    def heads(x):
        return x.view(B, N, H, D).permute(0, 2, 1, 3)
    oref, _ = _attn_ref(heads(qkv[:, :C]), heads(qkv[:, C:2 * C]), heads(qkv[:, 2 * C:]), D ** -0.5)
    assert relerr(o, oref.permute(0, 2, 1, 3).reshape(B * N, C)) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(100, 64), (513, 512), (64, 1024)])
def test_layernorm(dev, dtype, M, C):
    ops, L = _ops()
    g = torch.Generator().manual_seed(9)
    x, xf = rnd((M, C), dtype, dev, g); dy, dyf = rnd((M, C), dtype, dev, g); dres, dresf = rnd((M, C), dtype, dev, g)
    gamma = torch.randn(C, generator=g); beta = torch.randn(C, generator=g)
    xr = xf.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5); ref.backward(dyf)
    y = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    ops.layernorm_fwd(x, gamma.to(dev), beta.to(dev), y, mean, rstd, 1e-5)
    assert relerr(y, ref) < TOL[dtype]
    dx = torch.empty_like(x); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    ops.layernorm_bwd(dy, x, mean, rstd, gamma.to(dev), dres, dx, dg, db)
    assert relerr(dx, xr.grad + dresf) < TOL[dtype]
    assert relerr(dg, gr.grad) < TOL[dtype] and relerr(db, br.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,C1,C2,G,silu", [(2, 70, 64, 0, 32, True), (2, 64, 128, 64, 32, True), (3, 33, 256, 256, 32, False),
                                               (1, 130, 512, 512, 32, True)])
def test_groupnorm(dev, dtype, B, N, C1, C2, G, silu):
    ops, L = _ops()
    g = torch.Generator().manual_seed(10)
    C = C1 + C2
    x1, x1f = rnd((B * N, C1), dtype, dev, g)
    x2, x2f = rnd((B * N, C2), dtype, dev, g) if C2 else (None, None)
    dy, dyf = rnd((B * N, C), dtype, dev, g); dres, dresf = rnd((B * N, C), dtype, dev, g)
    gamma = 1 + 0.3 * torch.randn(C, generator=g); beta = 0.3 * torch.randn(C, generator=g)
    xcat = (torch.cat([x1f, x2f], 1) if C2 else x1f)
    xr = xcat.view(B, N, C).permute(0, 2, 1).clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    z = F.group_norm(xr, G, gr, br, 1e-5); ref = F.silu(z) if silu else z
    ref.backward(dyf.view(B, N, C).permute(0, 2, 1))
    mean = torch.empty(B * G, device=dev); rstd = torch.empty(B * G, device=dev)
    ops.groupnorm_stats(x1, x2, mean, rstd, B, N, G, 1e-5)
    y = torch.empty(B * N, C, dtype=dtype, device=dev); xc = torch.empty_like(y)
    ops.groupnorm_apply(x1, x2, mean, rstd, gamma.to(dev), beta.to(dev), y, xc, B, N, G, silu)
    assert relerr(y.view(B, N, C).permute(0, 2, 1), ref) < TOL[dtype]
    assert torch.equal(xc.float().cpu(), xcat)
    dx1 = torch.empty_like(x1); dx2 = torch.ones_like(x2) if C2 else None
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); ws = torch.empty(B * G * 2, device=dev)
    ops.groupnorm_bwd(dy, x1, x2, mean, rstd, gamma.to(dev), beta.to(dev), dres, dx1, dx2, dg, db, ws, B, N, G, silu,
                      accumulate_dx2=True)
    dxref = xr.grad.permute(0, 2, 1).reshape(B * N, C) + dresf
    assert relerr(dx1, dxref[:, :C1]) < TOL[dtype] * 2
    if C2:
        assert relerr(dx2, dxref[:, C1:] + 1.0) < TOL[dtype] * 2
    assert relerr(dg, gr.grad) < TOL[dtype] * 2 and relerr(db, br.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,C1,C2,G,silu", [(2, 1024, 512, 0, 32, True), (2, 96, 256, 256, 32, True), (3, 700, 512, 512, 32, True),
                                               (2, 70, 64, 0, 32, False), (1, 1030, 128, 0, 8, True), (2, 200, 512, 0, 8, True),
                                               (2, 64, 96, 32, 16, False),
                                               (2, 2048, 1024, 0, 32, True), (2, 1900, 512, 512, 32, True), (2, 2048, 1024, 1024, 32, True)])
def test_groupnorm_fused_forward_backward(dev, dtype, B, N, C1, C2, G, silu):
    """pt_groupnorm_fwd (one slab kernel for bf16: N <= 1024, or N <= 2048 with 32-channel slabs -- config E's items; the
    last case, 64 channels per group at N = 2048, takes the two-pass kernels) + pt_groupnorm_bwd against torch."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(13)
    C = C1 + C2
    x1, x1f = rnd((B * N, C1), dtype, dev, g)
    x2, x2f = rnd((B * N, C2), dtype, dev, g) if C2 else (None, None)
    dy, dyf = rnd((B * N, C), dtype, dev, g); dres, dresf = rnd((B * N, C), dtype, dev, g)
    gamma = 1 + 0.3 * torch.randn(C, generator=g); beta = 0.3 * torch.randn(C, generator=g)
    xcat = (torch.cat([x1f, x2f], 1) if C2 else x1f)
    xr = xcat.view(B, N, C).permute(0, 2, 1).clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    z = F.group_norm(xr, G, gr, br, 1e-5); ref = F.silu(z) if silu else z
    ref.backward(dyf.view(B, N, C).permute(0, 2, 1))
    mean = torch.empty(B * G, device=dev); rstd = torch.empty(B * G, device=dev)
    y = torch.empty(B * N, C, dtype=dtype, device=dev)
    ops.groupnorm_fwd(x1, x2, gamma.to(dev), beta.to(dev), y, mean, rstd, B, N, G, 1e-5, silu)
    assert relerr(y.view(B, N, C).permute(0, 2, 1), ref) < TOL[dtype]
    xg = xcat.view(B, N, G, C // G).permute(0, 2, 1, 3).reshape(B * G, -1)
    assert relerr(mean, xg.mean(1)) < 1e-4 + TOL[dtype] * 0 and relerr(rstd, (xg.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-4
    dx1 = torch.empty_like(x1); dx2 = torch.ones_like(x2) if C2 else None
    n_rep, C_al = 4, (C + 63) // 64 * 64
    rep = torch.zeros(2, n_rep, C_al, device=dev); ws = torch.empty(B * G * 2, device=dev)
    item_sum = torch.full((B, C + 8), 0.5, device=dev)[:, 4:4 + C] if not C2 else None      # a strided view, pre-loaded
    ops.groupnorm_bwd(dy, x1, x2, mean, rstd, gamma.to(dev), beta.to(dev), dres, dx1, dx2, rep[0, 0], rep[1, 0], ws, B, N, G,
                      silu, accumulate_dx2=True, n_rep=n_rep, rep_stride=C_al, item_sum=item_sum)
    dg = rep[0].sum(0)[:C]; db = rep[1].sum(0)[:C]
    dxref = xr.grad.permute(0, 2, 1).reshape(B * N, C) + dresf
    if item_sum is not None:          # per-item column sums of dx, accumulated into the destination (slab kernel / two-pass + colsum)
        assert relerr(item_sum - 0.5, dxref.view(B, N, C).sum(1)) < TOL[dtype] * 2
    assert relerr(dx1, dxref[:, :C1]) < TOL[dtype] * 2
    if C2:
        assert relerr(dx2, dxref[:, C1:] + 1.0) < TOL[dtype] * 2
    assert relerr(dg, gr.grad) < TOL[dtype] * 2 and relerr(db, br.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
def test_groupnorm_raw_statistics(dev, dtype):
    """Raw mode (sums in caller-zeroed scratch, finalized by the consumers) == finalized mode, bit for bit."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(12)
    B, N, C1, C2, G = 2, 96, 128, 64, 32
    C = C1 + C2
    x1, _ = rnd((B * N, C1), dtype, dev, g); x2, _ = rnd((B * N, C2), dtype, dev, g)
    dy, _ = rnd((B * N, C), dtype, dev, g)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).to(dev); beta = (0.3 * torch.randn(C, generator=g)).to(dev)
    outs = []
    for raw in (False, True):
        if raw:
            sums = torch.zeros(2 * B * G, device=dev); mean, rstd = sums[:B * G], sums[B * G:]
            ops.groupnorm_stats(x1, x2, mean, rstd, B, N, G, -1.0)
            ws = torch.zeros(B * G * 2, device=dev)
        else:
            mean = torch.empty(B * G, device=dev); rstd = torch.empty(B * G, device=dev)
            ops.groupnorm_stats(x1, x2, mean, rstd, B, N, G, 1e-5)
            ws = torch.empty(B * G * 2, device=dev)
        y = torch.empty(B * N, C, dtype=dtype, device=dev)
        ops.groupnorm_apply(x1, x2, mean, rstd, gamma, beta, y, None, B, N, G, True, raw_eps=1e-5 if raw else -1.0)
        dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2)
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        ops.groupnorm_bwd(dy, x1, x2, mean, rstd, gamma, beta, None, dx1, dx2, dg, db, ws, B, N, G, True,
                          raw_eps=1e-5 if raw else -1.0, ws_zeroed=raw)
        outs.append((y, dx1, dx2))
    for a, b in zip(*outs):                                  # statistics are atomics-ordered: tolerance, not equality
        assert relerr(a, b.float().cpu()) < 1e-5 * (1 if dtype == torch.float32 else 1000)


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise(dev, dtype):
    ops, L = _ops()
    g = torch.Generator().manual_seed(11)
    M, Fh = 130, 256
    proj, pf = rnd((M, 2 * Fh), dtype, dev, g); dout, df = rnd((M, Fh), dtype, dev, g)
    pr = pf.clone().requires_grad_(True)
    h, gate = pr.chunk(2, -1); ref = h * F.gelu(gate); ref.backward(df)
    out = torch.empty(M, Fh, dtype=dtype, device=dev); dproj = torch.empty_like(proj)
    ops.geglu_fwd(proj, out); ops.geglu_bwd(dout, proj, dproj)
    assert relerr(out, ref) < TOL[dtype] and relerr(dproj, pr.grad) < TOL[dtype]
    # interleaved column order (pt_gemm act 2 / 3 layout) + in-place bias: same function of permuted inputs
    idx = torch.arange(2 * Fh); q, t = idx // 64, idx % 64
    src = torch.where(t < 32, 32 * q + t, Fh + 32 * q + (t - 32))
    bias = torch.randn(2 * Fh, generator=g)
    pil = proj[:, src.to(dev)].contiguous(); out2 = torch.empty_like(out); dpil = torch.empty_like(pil)
    ops.geglu_fwd(pil, out2, bias=bias.to(dev), interleaved=True)
    pb = (pf + bias).clone().requires_grad_(True)
    hb, gb_ = pb.chunk(2, -1); refb = hb * F.gelu(gb_); refb.backward(df)
    assert relerr(out2, refb) < TOL[dtype] and relerr(pil, (pf + bias)[:, src]) < TOL[dtype]
    ops.geglu_bwd(dout, pil, dpil, interleaved=True)
    assert relerr(dpil, pb.grad[:, src]) < TOL[dtype] * 2
    x, xf = rnd((1000 + 3,), dtype, dev, g); dy, dyf = rnd((1003,), dtype, dev, g)
    xr = xf.clone().requires_grad_(True); F.silu(xr).backward(dyf)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    ops.silu_fwd(x, y); ops.silu_bwd(dy, x, dx)
    assert relerr(y, F.silu(xf)) < TOL[dtype] and relerr(dx, xr.grad) < TOL[dtype]
    s = torch.empty_like(x); ops.add(x, dy, s)
    assert relerr(s, xf + dyf) < TOL[dtype]
    a, af = rnd((64, 128), dtype, dev, g); ps = torch.empty(32, 128, dtype=dtype, device=dev)
    ops.pairsum_rows(a, ps)
    assert relerr(ps, af.view(32, 2, 128).sum(1)) < TOL[dtype]
    cs = torch.zeros(128, device=dev); ops.colsum(a, cs)
    assert relerr(cs, af.sum(0)) < TOL[dtype]
    seg = torch.zeros(4, 200, device=dev)                 # segmented sums into a column slice of a wider matrix
    ops.colsum(a, seg[:, 40:], 64, 128, seg_rows=16, ld_out=200)
    assert relerr(seg[:, 40:168], af.view(4, 16, 128).sum(1)) < TOL[dtype] and float(seg[:, :40].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding_timestep_noise_mse(dev, dtype):
    ops, L = _ops()
    g = torch.Generator().manual_seed(12)
    B, S, d, V = 3, 40, 64, 149
    ids = torch.randint(0, V, (B, S), generator=g, dtype=torch.int32)
    W, Wf = rnd((V, d), dtype, dev, g); pos = torch.randn(S, d, generator=g)
    out = torch.empty(B * S, d, dtype=dtype, device=dev)
    ops.embedding_fwd(ids.to(dev), W, pos.to(dev), out, S)
    ref = Wf[ids.long()] + pos[None]
    assert relerr(out, ref.view(B * S, d)) < TOL[dtype]
    dout, df = rnd((B * S, d), dtype, dev, g); dW = torch.zeros(V, d, device=dev)
    ops.embedding_bwd(ids.to(dev).view(-1), dout, dW)
    refW = torch.zeros(V, d).index_add_(0, ids.view(-1).long(), df)
    assert relerr(dW, refW) < TOL[dtype]

    from oracle.blocks import timestep_embedding, add_noise, ddpm_alphas_cumprod   # oracle = checker only
    t = torch.tensor([0, 1, 500, 999]); te = torch.empty(4, 256, dtype=dtype, device=dev)
    ops.timestep_embedding(t.to(dev), te)
    assert relerr(te, timestep_embedding(t, 256)) < max(TOL[dtype], 2e-4)

    n_q, T, cpad = 2, 50, 8
    x0 = torch.rand(4, n_q, T, generator=g) * 2 - 1; noise = torch.randn(4, n_q, T, generator=g)
    ac = ddpm_alphas_cumprod()
    xt = torch.empty(4 * T, cpad, dtype=dtype, device=dev)
    ops.add_noise(x0.to(dev), noise.to(dev), t.to(dev), ac.to(dev), xt, n_q, T, cpad)
    ref = add_noise(x0, noise, t, ac)
    back = torch.empty(4, n_q, T, device=dev); ops.tokens_to_bct(xt, back, 4, n_q, T, cpad)
    assert relerr(back, ref) < TOL[dtype]
    assert float(xt[:, n_q:].float().abs().max()) == 0.0
    pred, pf = rnd((4 * T, cpad), dtype, dev, g)
    loss = torch.zeros(1, device=dev); dpred = torch.empty_like(pred)
    ops.mse_loss(pred, noise.to(dev), loss, dpred, 1.0, 4, n_q, T, cpad)
    pbct = pf.view(4, T, cpad)[:, :, :n_q].permute(0, 2, 1).clone().requires_grad_(True)
    lref = F.mse_loss(pbct, noise); lref.backward()
    assert abs(float(loss) - float(lref)) < 1e-5 * max(1.0, float(lref)) * (1 if dtype == torch.float32 else 10)
    assert relerr(dpred.view(4, T, cpad)[:, :, :n_q].permute(0, 2, 1), pbct.grad) < TOL[dtype]
