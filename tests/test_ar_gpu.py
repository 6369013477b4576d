"""GPU: the north_star ops with no reference counterpart (SURVEY 8a'): RVQ-codebook logits head, the autoregressive decode
loop with its K/V cache, causal self-attention + cross-attention blocks.  Build-defined; the oracle is plain PyTorch
(oracle/ar.py, torch.argmax / torch.topk through oracle/collate.sample_topk) -- parity unpinned by the reference by
construction.  Tolerances: f32 mode 1e-3 on logits (north_star), bf16 4e-2; indices bit-exact except where the oracle's own
top-2 logits are closer than the logit tolerance (a tie the two summation orders may break differently)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(got, ref):
    got = got.detach().float().cpu(); ref = ref.detach().float().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1.5e-2)])
def test_logits_head_at_configs3_size(dev, dtype, tol):
    """BASELINE configs[3]: 64 prompts x 1024 frames of d_model 512 -> 8 codebooks x 1024 logits each (65 536 x 512 -> 8192)."""
    from prompt_tts_amd import engine as E
    from prompt_tts_amd.ar import LogitsHead
    g = torch.Generator(device=dev).manual_seed(3)
    head = LogitsHead(512, 8, 1024).to(dev)
    st = E.ParamStore(head, dev, dtype)
    rows = 65536 if dtype == torch.bfloat16 else 8192
    h = (torch.randn(rows, 512, generator=g, device=dev) * 0.5).to(dtype)
    logits = head.fwd(st, h)
    assert logits.shape == (rows, 8192) and logits.dtype == dtype
    w = head.proj.weight.detach().to(dtype).float(); b = head.proj.bias.detach().float()
    ref = h.float() @ w.t() + b
    assert relerr(logits, ref) < tol
    # greedy tokens from those logits: bit-exact against torch.argmax of the SAME logits
    from prompt_tts_amd import ops
    idx = ops.sample_topk(logits.view(rows * 8, 1024), 1)
    assert torch.equal(idx, logits.view(rows * 8, 1024).float().argmax(-1))


def _pair(dev, dtype, seed=0, d=256, L=2, n_q=4, bins=128, heads=4, T=96):
    from oracle import ar as oar
    from oracle.init import deterministic_init_
    from prompt_tts_amd.ar import ARCodecDecoder
    ref = deterministic_init_(oar.ARCodecDecoder(d, L, n_q, bins, heads, max_frames=T), seed + 1)
    with torch.no_grad():
        ref.code_embedding.mul_(20.0); ref.bos.mul_(5.0)                     # O(1) inputs
    m = ARCodecDecoder(d, L, n_q, bins, heads, max_frames=T, dtype=dtype)
    assert list(m.state_dict()) == list(ref.state_dict())
    m.load_state_dict(ref.state_dict())
    return ref, m.to(dev)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
def test_teacher_forced_logits_vs_oracle_and_causality(dev, dtype, tol):
    ref, m = _pair(dev, dtype)
    g = torch.Generator().manual_seed(1)
    B, T, S = 3, 96, 40
    codes = torch.randint(0, 128, (B, 4, T), generator=g); ctx = torch.randn(B, S, 256, generator=g)
    with torch.no_grad():
        want = ref(codes, ctx)
    got = m(codes.to(dev), ctx.to(dev))
    assert got.shape == (B, T, 4, 128) and relerr(got, want) < tol
    # causal: frames >= 50 changed -> logits of positions <= 50 unchanged (position t sees frames < t)
    codes2 = codes.clone(); codes2[:, :, 50:] = torch.randint(0, 128, (B, 4, T - 50), generator=g)
    got2 = m(codes2.to(dev), ctx.to(dev))
    assert relerr(got2[:, :51], got[:, :51]) < (1e-5 if dtype == torch.float32 else 1e-2) and relerr(got2[:, 60:], got[:, 60:]) > 1e-3


@pytest.mark.parametrize("k", [1, 8])
def test_generate_with_kv_cache_equals_teacher_forced_decisions(dev, k):
    """The AR loop (one frame per step, K/V cache, cross K/V projected once) must make, at every step, the decision the
    oracle makes from the teacher-forced logits of the sequence generated so far -- the cache is exact, not an approximation."""
    from oracle import collate as oc
    ref, m = _pair(dev, torch.float32, seed=3)
    g = torch.Generator().manual_seed(2)
    B, T, S = 4, 64, 24
    ctx = torch.randn(B, S, 256, generator=g)
    u = torch.rand(T, B * 4, generator=g)
    codes = m.generate(ctx.to(dev), T, k=k, uniforms=u.to(dev) if k > 1 else None, temperature=0.9).cpu()
    assert codes.shape == (B, 4, T) and int(codes.min()) >= 0 and int(codes.max()) < 128 and len(torch.unique(codes)) > 20
    with torch.no_grad():
        logits = ref(codes, ctx)                                                   # oracle logits of the generated sequence
    bad = 0
    for t in range(T):
        lt = logits[:, t].reshape(B * 4, 128)
        want = oc.sample_topk(lt, k, u[t] if k > 1 else None, temperature=0.9) if k > 1 else lt.argmax(-1)
        miss = want != codes[:, :, t].reshape(-1)
        if miss.any():                                                             # only ties within the logit tolerance may differ
            top2 = lt.topk(min(k + 1, 128), dim=-1).values
            near = (top2[:, :-1] - top2[:, 1:]).abs().min(-1).values < 1e-3 * float(lt.abs().max())
            bad += int((miss & ~near).sum())
            if k > 1:                                                              # or a uniform within tolerance of a CDF edge
                bad -= int((miss & ~near).sum()); bad += int(miss.sum() > 2)
    assert bad == 0
    # the device's own teacher-forced pass agrees with its incremental pass
    lg = m(codes.to(dev), ctx.to(dev)).cpu()
    assert relerr(lg, logits) < 1e-3


@pytest.mark.parametrize("k,dtype", [(1, torch.float32), (8, torch.bfloat16)])
def test_generate_graph_replay_equals_launch_by_launch(dev, k, dtype):
    """generate() captures ONE decode step (frame index, K/V length, previous codes, uniforms row on the device) as a HIP graph and
    replays it; launch by launch (graph=False) the same kernels run in the same order: the codes are identical."""
    _, m = _pair(dev, dtype, seed=5)
    g = torch.Generator().manual_seed(7)
    B, T, S = 3, 40, 24
    ctx = torch.randn(B, S, 256, generator=g).to(dev)
    u = torch.rand(T, B * 4, generator=g).to(dev)
    a = m.generate(ctx, T, k=k, uniforms=u if k > 1 else None, temperature=0.8, graph=True)
    b = m.generate(ctx, T, k=k, uniforms=u if k > 1 else None, temperature=0.8, graph=False)
    assert a.shape == (B, 4, T) and torch.equal(a, b) and len(torch.unique(a)) > 10
    c = m.generate(ctx, T, k=k, uniforms=u if k > 1 else None, temperature=0.8, graph=True)      # a second capture in the same process
    assert torch.equal(a, c)


@pytest.mark.parametrize("M,K,N,mode", [(64, 512, 1536, "qkv"), (37, 256, 256, "plain"), (64, 2048, 512, "res"), (5, 512, 4096, "geglu"),
                                        (64, 512, 8192, "ln_bias"), (64, 512, 512, "res"), (3, 1024, 256, "res"), (64, 256, 768, "qkv"),
                                        (64, 256, 2048, "geglu"), (64, 512, 512, "ln_bias")])
def test_decode_linear_equals_the_separate_launches_bit_for_bit(dev, M, K, N, mode):
    """pt_decode_linear ([LayerNorm ->] Linear [-> bias] [-> + residual] [-> GEGLU] [-> K / V cache scatter], M <= 64 rows, one
    launch) against the launches it folds -- pt_layernorm_fwd, pt_gemm, pt_geglu_fwd, index_copy_ -- on the same inputs: the SAME
    bits (same rounding points, same accumulation order), and a torch f32 reference within bf16 tolerance."""
    from prompt_tts_amd import engine as E, ops
    g = torch.Generator().manual_seed(M + K + N)
    bf = torch.bfloat16
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).to(dev, bf)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev, bf)
    bias = torch.randn(N, generator=g).to(dev)
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(dev); beta = (0.1 * torch.randn(K, generator=g)).to(dev)
    if mode == "qkv":
        C, Tm = N // 3, 7
        kc = torch.zeros(M, Tm, C, device=dev, dtype=bf); vc = torch.zeros_like(kc)
        kc2, vc2 = kc.clone(), vc.clone()
        t_dev = torch.full((1,), 4, dtype=torch.int64, device=dev)
        q = torch.empty(M, C, device=dev, dtype=bf)
        ops.decode_linear(x, w, q, M, N, K, ln=(gamma, beta), seg_cols=C, y2=kc, ld2=Tm * C, y3=vc, ld3=Tm * C, t_dev=t_dev, t_stride=C)
        n1, _ = E.layernorm_fwd(x, gamma, beta)
        qkv = E.linear_fwd(n1, w)
        kc2.index_copy_(1, t_dev, qkv[:, C:2 * C].unsqueeze(1)); vc2.index_copy_(1, t_dev, qkv[:, 2 * C:].unsqueeze(1))
        assert torch.equal(q, qkv[:, :C]) and torch.equal(kc, kc2) and torch.equal(vc, vc2)
        ref = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta) @ w.float().t()
        assert relerr(q, ref[:, :C]) < 3e-2
    elif mode in ("plain", "res"):
        res = (torch.randn(M, N, generator=g)).to(dev, bf) if mode == "res" else None
        y = torch.empty(M, N, device=dev, dtype=bf)
        ops.decode_linear(x, w, y, M, N, K, bias=bias, residual=res)
        want = E.linear_fwd(x, w, bias, residual=res)
        assert torch.equal(y, want)
        ref = x.float() @ w.float().t() + bias + (res.float() if res is not None else 0)
        assert relerr(y, ref) < 2e-2
    elif mode == "ln_bias":
        y = torch.empty(M, N, device=dev, dtype=bf)
        ops.decode_linear(x, w, y, M, N, K, ln=(gamma, beta), bias=bias)
        n1, _ = E.layernorm_fwd(x, gamma, beta)
        assert torch.equal(y, E.linear_fwd(n1, w, bias))
    else:
        # GEGLU: w rows in the interleaved shadow order (value row 32 q + t at 64 q + t, its gate at 64 q + 32 + t), bias in the original order
        F = N // 2
        perm = torch.empty(N, dtype=torch.int64)
        for mi in range(N):
            qq, t = mi >> 6, mi & 63
            perm[mi] = 32 * qq + t if t < 32 else F + 32 * qq + (t - 32)
        wi = w[perm.to(dev)].contiguous()
        act = torch.empty(M, F, device=dev, dtype=bf)
        ops.decode_linear(x, wi, act, M, N, K, ln=(gamma, beta), bias=bias, geglu=True)
        n1, _ = E.layernorm_fwd(x, gamma, beta)
        proj = E.linear_fwd(n1, wi)
        want = torch.empty(M, F, device=dev, dtype=bf)
        ops.geglu_fwd(proj, want, bias=bias, interleaved=True)
        assert torch.equal(act, want)
        pr = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta) @ w.float().t() + bias
        ref = pr[:, :F] * torch.nn.functional.gelu(pr[:, F:])
        assert relerr(act, ref) < 3e-2


@pytest.mark.parametrize("K", [256, 512])
def test_decode_linear_layernorm_is_pt_layernorm_fwd_bit_for_bit(dev, K):
    """The LayerNorm prologue of pt_decode_linear (through an identity weight) against pt_layernorm_fwd on 200 x 64 rows of varied
    scale and offset: EVERY element the same bf16 bits.  (Compiled from the same source expression the two kernels once differed
    in 1 element of ~30 000 -- the backend fused multiply-adds differently -- which flipped sampler near-ties a few frames on.)"""
    from prompt_tts_amd import engine as E, ops
    g = torch.Generator(device=dev).manual_seed(K)
    eye = torch.eye(K, device=dev, dtype=torch.bfloat16)
    gamma = (1 + 0.3 * torch.randn(K, generator=g, device=dev)); beta = 0.2 * torch.randn(K, generator=g, device=dev)
    bad = 0
    for trial in range(200):
        x = (torch.randn(64, K, generator=g, device=dev) * (0.05 + 0.1 * trial) + 0.01 * trial).to(torch.bfloat16)
        ref, _ = E.layernorm_fwd(x, gamma, beta)
        got = torch.empty_like(x)
        ops.decode_linear(x, eye, got, 64, K, K, ln=(gamma, beta))
        bad += int((ref != got).sum())
    assert bad == 0, bad


@pytest.mark.parametrize("k", [1, 8])
def test_folded_decode_step_makes_the_training_kernels_decisions(dev, k, monkeypatch):
    """generate() on the folded launches of csrc/decode_step.hip (~37 per frame) against the same loop on the training kernels
    (PT_AR_FUSED=0, ~70 per frame): the codes must be IDENTICAL -- the folded kernels keep every rounding point and accumulation
    order of the launches they replace."""
    import prompt_tts_amd.ar as par
    _, m = _pair(dev, torch.bfloat16, seed=9)
    g = torch.Generator().manual_seed(11)
    B, T, S = 5, 48, 24
    ctx = torch.randn(B, S, 256, generator=g).to(dev)
    u = torch.rand(T, B * 4, generator=g).to(dev)
    assert m._fusable(m.store, B)
    a = m.generate(ctx, T, k=k, uniforms=u if k > 1 else None, temperature=0.8, graph=True)
    b = m.generate(ctx, T, k=k, uniforms=u if k > 1 else None, temperature=0.8, graph=False)
    monkeypatch.setattr(par, "AR_FUSED", False)
    assert not m._fusable(m.store, B)
    c = m.generate(ctx, T, k=k, uniforms=u if k > 1 else None, temperature=0.8, graph=False)
    assert torch.equal(a, b) and torch.equal(a, c) and len(torch.unique(a)) > 10
