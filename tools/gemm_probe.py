#!/usr/bin/env python
"""Ablation probe for the GEMM kernel (run on the GPU box): builds variants of gemm.hip with parts of the kernel
compiled out and times representative shapes of the training step.  Diagnostic only; never imported by the product."""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prompt_tts_amd import _lib as L   # noqa: E402  (struct definitions)

SRC = os.path.join(ROOT, "prompt_tts_amd", "csrc")
SHAPES = [("lin K512 N512", 32768, 512, 512), ("lin K512 N1536", 32768, 1536, 512), ("ff1 K512 N4096", 32768, 4096, 512),
          ("ff2 K2048 N512", 32768, 512, 2048), ("conv K1536 N512", 32768, 512, 1536)]


def build(ablate):
    out = f"/tmp/libgemm_ab{ablate}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    f"-DPT_GEMM_ABLATE={0 if ablate >= 7 else ablate}", f"-DPT_EPI_NT={1 if ablate == 7 else (0 if ablate == 8 else -1)}",
                    os.path.join(SRC, "gemm.hip"), os.path.join(SRC, "capi.hip"), "-o", out], check=True)
    lib = C.CDLL(out)
    lib.pt_gemm.argtypes = [C.POINTER(L.pt_gemm_desc), C.c_int, C.c_void_p]
    lib.pt_gemm.restype = C.c_int
    return lib


NBUF = int(os.environ.get("PT_PROBE_NBUF", "1"))      # > 1: rotate over that many operand / output sets (defeats the 256 MB L3)


def run(lib, M, N, K, iters=30):
    descs = []; keep = []
    for _ in range(NBUF):
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        d = L.pt_gemm_desc(); d.M, d.N, d.K = M, N, K
        d.A.p, d.A.ld = a.data_ptr(), K; d.B.p, d.B.ld = w.data_ptr(), K
        d.C, d.ldc, d.split_k, d.alpha = c.data_ptr(), N, 1, 1.0
        descs.append(d); keep.append((a, w, c))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(3):
        assert lib.pt_gemm(C.byref(descs[i % NBUF]), 1, st) == 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        lib.pt_gemm(C.byref(descs[i % NBUF]), 1, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run_mode(lib, mode, tokens, n_out, k_in, split_k=1, iters=30):
    """mode 'dgrad': dx[tokens,k_in] = dy[tokens,n_out] W[n_out,k_in];  'wgrad': gw[n_out,k_in] += dy^T x (f32 atomics)."""
    dy = torch.randn(tokens, n_out, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(tokens, k_in, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n_out, k_in, device="cuda", dtype=torch.bfloat16)
    d = L.pt_gemm_desc()
    if mode == "dgrad":
        out = torch.empty(tokens, k_in, device="cuda", dtype=torch.bfloat16)
        d.M, d.N, d.K = tokens, k_in, n_out
        d.A.p, d.A.ld = dy.data_ptr(), n_out
        d.B.p, d.B.ld, d.B.trans = w.data_ptr(), k_in, 1
        d.C, d.ldc, d.split_k, d.alpha = out.data_ptr(), k_in, 1, 1.0
    else:
        out = torch.zeros(n_out, k_in, device="cuda", dtype=torch.float32)
        d.M, d.N, d.K = n_out, k_in, tokens
        d.A.p, d.A.ld, d.A.trans = dy.data_ptr(), n_out, 1
        d.B.p, d.B.ld, d.B.trans = x.data_ptr(), k_in, 1
        d.C, d.ldc, d.split_k, d.alpha, d.out_kind = out.data_ptr(), k_in, split_k, 1.0, L.PT_OUT_F32_ATOMIC
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.pt_gemm(C.byref(d), 1, st) == 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.pt_gemm(C.byref(d), 1, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


BWD_SHAPES = [("lin 512<-512", 32768, 512, 512), ("qkv 1536<-512", 32768, 1536, 512), ("ff1 4096<-512", 32768, 4096, 512),
              ("ff2 512<-2048", 32768, 512, 2048), ("lin T8192", 8192, 512, 512)]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--bwd":       # dgrad / wgrad shapes on the product library
        lib = C.CDLL(L.LIB_PATH)
        lib.pt_gemm.argtypes = [C.POINTER(L.pt_gemm_desc), C.c_int, C.c_void_p]; lib.pt_gemm.restype = C.c_int
        import math
        for name, T, NO, KI in BWD_SHAPES:
            fl = 2.0 * T * NO * KI
            us = run_mode(lib, "dgrad", T, NO, KI)
            row = f"{name:16s} dgrad {us:8.1f} us {fl / us / 1e6:6.0f} TF |"
            tiles = math.ceil(NO / 128) * math.ceil(KI / 128)
            for tgt in (128, 256, 512, 1024):
                sk = max(1, min(tgt // tiles, T // 64 // 4, 64))
                us = run_mode(lib, "wgrad", T, NO, KI, sk)
                row += f" wgrad sk{sk:<3d} {us:7.1f} us {fl / us / 1e6:5.0f} TF |"
            print(row, flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--pmc":      # one shape on the product library, for rocprofv3 --pmc runs
        lib = C.CDLL(L.LIB_PATH)
        lib.pt_gemm.argtypes = [C.POINTER(L.pt_gemm_desc), C.c_int, C.c_void_p]; lib.pt_gemm.restype = C.c_int
        name, M, N, K = SHAPES[int(sys.argv[2])]
        print(name, run(lib, M, N, K, iters=10), "us")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--bwd-ab":    # ablation variants on the wgrad shapes (split-K sized for 512 workgroups)
        import math
        variants = [int(x) for x in sys.argv[2:]] or [0, 1, 2, 3]
        libs = {v: build(v) for v in variants}
        print("wgrad shape".ljust(22) + "".join(f"ab{v}:us/TF".rjust(18) for v in variants), flush=True)
        for name, T, NO, KI in BWD_SHAPES:
            tiles = math.ceil(NO / 128) * math.ceil(KI / 128)
            sk = max(1, min(512 // tiles, T // 64 // 4, 64))
            row = f"{name} sk{sk}".ljust(22)
            for v in variants:
                us = run_mode(libs[v], "wgrad", T, NO, KI, sk)
                row += f"{us:9.1f}/{2.0 * T * NO * KI / us / 1e6:6.0f}".rjust(18)
            print(row, flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--trace":    # s_memtime stamps of wave 0 / workgroup 0 (100 MHz counter: 10 ns units)
        out = "/tmp/libgemm_trace.so"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPT_GEMM_TRACE=1",
                        os.path.join(SRC, "gemm.hip"), os.path.join(SRC, "capi.hip"), "-o", out], check=True)
        lib = C.CDLL(out)
        lib.pt_gemm.argtypes = [C.POINTER(L.pt_gemm_desc), C.c_int, C.c_void_p]; lib.pt_gemm.restype = C.c_int
        buf = (C.c_ulonglong * 8)()
        for name, M, N, K in SHAPES[:3]:
            run(lib, M, N, K, iters=5)
            torch.cuda.synchronize()
            lib.pt_debug_gemm_trace(buf)
            t = [int(buf[i]) for i in range(6)]
            print(name, "stamps (x10 ns) relative to entry:", [x - t[0] for x in t], flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--ktrace":   # per k-tile stamps of the two-stage 256 x 256 kernel (PT_GEMM_TILE=512)
        os.environ["PT_GEMM_TILE"] = os.environ.get("PT_GEMM_TILE", "512")
        out = "/tmp/libgemm_trace.so"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPT_GEMM_TRACE=1",
                        os.path.join(SRC, "gemm.hip"), os.path.join(SRC, "capi.hip"), "-o", out], check=True)
        lib = C.CDLL(out)
        lib.pt_gemm.argtypes = [C.POINTER(L.pt_gemm_desc), C.c_int, C.c_void_p]; lib.pt_gemm.restype = C.c_int
        buf = (C.c_ulonglong * 136)()
        for name, M, N, K in SHAPES:
            us = run(lib, M, N, K, iters=200)          # long enough for the clock to settle
            torch.cuda.synchronize()
            lib.pt_debug_gemm_ktrace(buf)
            t = [int(x) for x in buf]
            nkt = min(K // 64, 32)
            rows = [t[8 + 4 * k:8 + 4 * k + 4] for k in range(nkt)]
            d = [[r[1] - r[0], r[2] - r[1], r[3] - r[2], (rows[k + 1][0] - r[3]) if k + 1 < nkt else 0] for k, r in enumerate(rows)]
            mid = d[1:-1] or d
            mean = [sum(x[i] for x in mid) / len(mid) for i in range(4)]
            ghz = (t[5] - t[0]) / max(t[7] - t[6], 1) * 0.1        # shader cycles per 100 MHz tick over the life of workgroup 0
            print(f"{name}: {us:.1f} us at {ghz:.2f} GHz (s_memtime / s_memrealtime); prologue {t[2] - t[0]} cycles, k-loop {rows[-1][3] - rows[0][0]}, epilogue {t[5] - t[3]}; per k-tile (mean of the inner ones): "
                  f"issue loads {mean[0]:.0f}, reads + MFMAs issued {mean[1]:.0f}, wait for the loads {mean[2]:.0f}, barrier {mean[3]:.0f}", flush=True)
            print("   first tiles:", d[:4], flush=True)
        sys.exit(0)
    variants = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3]
    libs = {v: build(v) for v in variants}
    print("shape".ljust(18) + "".join(f"ab{v}:us/TF".rjust(18) for v in variants), flush=True)
    for name, M, N, K in SHAPES:
        row = name.ljust(18)
        for v in variants:
            us = run(libs[v], M, N, K)
            row += f"{us:9.1f}/{2.0 * M * N * K / us / 1e6:6.0f}".rjust(18)
        print(row, flush=True)
