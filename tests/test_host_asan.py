"""SURVEY 5 / VERDICT r02 item 9: the HOST side of the C-ABI library under AddressSanitizer + UndefinedBehaviorSanitizer.

Every entry point's argument checks, descriptor walks, tile / grid / workspace arithmetic and launch-parameter packing are host
C++.  On this CPU box a launch fails with "no ROCm-capable device" AFTER all of that has run, so valid descriptors exercise the
whole host path (status PT_ERR_LAUNCH) without a GPU.  `make asan` (prompt_tts_amd/csrc) builds the same sources with
-fsanitize=address,undefined on the host side; the cases below and the refuse-before-launch cases of tests/test_host_cpu.py are
re-run against it in a child process with the ASan runtime preloaded.  (Device ASan needs xnack+, which the pool does not offer.)
"""
import glob
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "prompt_tts_amd", "csrc")
ASAN_LIB = os.path.join(CSRC, "build", "asan", "libprompt_tts_hip_asan.so")


def _no_gpu():
    return not torch.cuda.is_available()


@pytest.fixture()
def host_only(monkeypatch):
    if not _no_gpu():
        pytest.skip("host-path cases pass host pointers: only where no launch can succeed")
    from prompt_tts_amd import ops
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "_dev", lambda *ts: None)
    return ops


def _launch_fails(fn, *a, **k):
    with pytest.raises(RuntimeError, match="HIP launch failed"):
        fn(*a, **k)


def test_host_paths_reach_the_launch(host_only):
    ops = host_only
    from prompt_tts_amd import _lib as L
    bf, f32 = torch.bfloat16, torch.float32
    z = lambda *s, dt=bf: torch.zeros(*s, dtype=dt)
    # GEMM family: every tile choice (128^2, two-stage 256^2, eight-phase), f32, conv / concat / flipped operands, epilogues
    for M, N, K in ((300, 200, 72), (8192, 8192, 512), (8192, 8192, 2048), (256, 4096, 512)):
        a, b, c = z(M, K), z(N, K), z(M, N)
        _launch_fails(ops.gemm, M, N, K, ops.plain(a), ops.plain(b), c, L.PT_BF16, bias=z(N, dt=f32), residual=c, ldr=N)
    a, b, c = z(512, 64, dt=f32), z(96, 64, dt=f32), z(512, 96, dt=f32)
    _launch_fails(ops.gemm, 512, 96, 64, ops.plain(a), ops.plain(b), c, L.PT_F32)
    x, w3, y = z(4 * 64, 128), z(256, 3 * 128), z(4 * 64, 256)
    _launch_fails(ops.gemm, 256, 256, 384, ops.conv(x, 128, 64, 64, L.PT_MAP_S1), ops.plain(w3), y, L.PT_BF16)
    _launch_fails(ops.gemm, 128, 256, 384, ops.conv(x, 128, 32, 64, L.PT_MAP_S2), ops.plain(w3), z(128, 256), L.PT_BF16)
    _launch_fails(ops.gemm, 256, 128, 768, ops.conv(y, 256, 64, 64, L.PT_MAP_S1), ops.wflip(z(256, 3 * 128), 256, 128), z(256, 128), L.PT_BF16)
    _launch_fails(ops.gemm, 256, 128, 384, ops.concat(z(256, 128), z(256, 256)), ops.plain(z(128, 384)), z(256, 128), L.PT_BF16)
    proj, act = z(512, 4096), z(512, 2048)
    _launch_fails(ops.gemm, 512, 4096, 512, ops.plain(z(512, 512)), ops.plain(z(4096, 512)), proj, L.PT_BF16, bias=z(4096, dt=f32),
                  act=2, out2=act, ldc2=2048)
    # weight gradients: split-K atomics and the grouped launch (3 problems, bias gradient riding along)
    dy, xx, gw = z(4096, 512), z(4096, 256), z(512, 256, dt=f32)
    _launch_fails(ops.gemm, 512, 256, 4096, ops.plain(dy, trans=True), ops.plain(xx, trans=True), gw, L.PT_BF16, ldc=256,
                  out_kind=L.PT_OUT_F32_ATOMIC, split_k=8)
    descs = [ops.gemm_desc(512, 256, 4096, ops.plain(dy, trans=True), ops.plain(xx, trans=True), gw, ldc=256,
                           out_kind=L.PT_OUT_F32_ATOMIC, arow_sum=z(512, dt=f32), arow_n=512) for _ in range(3)]
    ws = torch.zeros(ops.wgrad_group_ws_floats(256) // 64, dtype=f32)              # the size check comes before any launch
    with pytest.raises(RuntimeError):
        ops.wgrad_group(descs, ws, 256)
    # attention, norms, elementwise, optimizer, Encodec
    B, H, N, D = 2, 4, 128, 64
    q = z(B * N, 3 * H * D); o = z(B * N, H * D); lse = z(B, H, N, dt=f32)
    _launch_fails(ops.attn_fwd, q[:, :256], q[:, 256:512], q[:, 512:], o, lse, B, H, N, N, D, 0.125)
    _launch_fails(ops.attn_fwd, q[:, :256], q[:, 256:512], q[:, 512:], o, lse, B, H, N, N, D, 0.125, True, torch.full((B,), 100, dtype=torch.int32))
    _launch_fails(ops.attn_bwd, q[:, :256], q[:, 256:512], q[:, 512:], o, lse, o, torch.zeros_like(lse), z(B * N, 256), z(B * N, 256),
                  z(B * N, 256), B, H, N, N, D, 0.125)
    xln = z(256, 512)
    _launch_fails(ops.layernorm_fwd, xln, z(512, dt=f32), z(512, dt=f32), z(256, 512), z(256, dt=f32), z(256, dt=f32))
    _launch_fails(ops.groupnorm_fwd, z(2 * 64, 256), None, z(256, dt=f32), z(256, dt=f32), z(2 * 64, 256), z(64, dt=f32), z(64, dt=f32),
                  2, 64, 32, 1e-5, True)
    _launch_fails(ops.groupnorm_stats, z(2 * 64, 256, dt=f32), z(2 * 64, 256, dt=f32), z(64, dt=f32), z(64, dt=f32), 2, 64, 32, 1e-5)
    _launch_fails(ops.geglu_fwd, proj, act, bias=z(4096, dt=f32), interleaved=True)
    _launch_fails(ops.sumsq, z(1000, dt=f32), z(1, dt=f32))
    st = z(258, dt=f32)
    _launch_fails(ops.fp8_quantize, z(256, 128), torch.zeros(256, 128, dtype=torch.uint8), st, 0, torch.zeros(128, 256, dtype=torch.uint8))
    _launch_fails(ops.gemm_fp8, 512, 512, 256, torch.zeros(512, 256, dtype=torch.uint8), torch.zeros(512, 256, dtype=torch.uint8),
                  z(512, 512), st, st)
    _launch_fails(ops.rvq_decode, torch.zeros(2, 8, 16, dtype=torch.int64), z(8, 1024, 128), z(32, 128), 2, 8, 16, 1024, 128)
    _launch_fails(ops.sample_topk, z(64, 1024), 32, z(64, dt=f32))
    from prompt_tts_amd import encodec as pe
    with pytest.raises(RuntimeError, match="pt_lstm2_forward"):       # per-step form (no device: 0 CUs) -> the memset / launch fails
        pe.run_lstm2(20, 150, z(3000, 512), z(3000, 2048), z(2048, 512), z(2048, 1024), z(2048, dt=f32), L.PT_BF16, "cpu", bf,
                     status=type("S", (), {"ptr": lambda s: z(1, dt=torch.int32).data_ptr(), "fetch": lambda s: None})())


def test_host_side_under_asan_and_ubsan():
    if not _no_gpu():
        pytest.skip("CPU-box check")
    if os.environ.get("PT_SKIP_ASAN") == "1":
        pytest.skip("PT_SKIP_ASAN=1")
    r = subprocess.run(["make", "-C", CSRC, "-j8", "asan"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert rt and os.path.exists(ASAN_LIB)
    syms = subprocess.run(["nm", "-D", ASAN_LIB], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms            # instrumented, not a plain rebuild
    env = dict(os.environ, LD_PRELOAD=rt[-1], PT_TTS_LIB=ASAN_LIB, PT_SKIP_ASAN="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host_cpu.py"),
                        os.path.join(ROOT, "tests", "test_host_asan.py"), "-k", "cabi or reach_the_launch"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0 and "3 passed" in p.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
