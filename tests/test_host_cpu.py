"""CPU: host logic -- collate parity, C-ABI surface, product/oracle separation, model construction, LR schedule."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _items(rng, n, n_q=2, T=12, with_norm=False, lens=(0, 3, 40)):
    out = []
    for i in range(n):
        L = lens[i % len(lens)]
        it = {"code": rng.integers(0, 1024, (n_q, T)).astype(np.int64) / 1023, "text": f"t{i}",
              "cmu_sequence": rng.integers(1, 149, L).tolist(), "code_length": float(T - i)}
        if with_norm:
            it["text_norm"] = f"n{i}"
        out.append(it)
    return out


@pytest.mark.parametrize("with_norm", [False, True])
def test_collate_bit_exact_vs_oracle(with_norm):
    from oracle import collate as oc
    from prompt_tts_amd.tts.dataloader import TTS_SingleSpkr_Collate_Fn, intersperse
    rng = np.random.default_rng(0)
    items = _items(rng, 5, with_norm=with_norm)              # empty, short and over-long (truncated) sequences
    got = TTS_SingleSpkr_Collate_Fn(16)(items)
    want = oc.collate([dict(it, code_int=np.rint(it["code"] * 1023).astype(np.int64)) for it in items], 16)
    assert got["cmu_sequence_id"].dtype == torch.int32 and got["attention_mask"].dtype == torch.int32
    assert np.array_equal(got["cmu_sequence_id"].numpy(), want["cmu_sequence_id"])            # bit-exact ints
    assert np.array_equal(got["attention_mask"].numpy(), want["attention_mask"])
    assert got["code"].dtype == torch.float32 and np.array_equal(got["code"].numpy(), want["code"])  # bit-exact f32
    assert set(got) == set(want) and ("text_norm" in got) == with_norm
    assert got["attention_mask"][0].sum() == 0 and got["attention_mask"][2].sum() == 16
    assert intersperse([5, 6], 148) == [148, 5, 148, 6, 148] == oc.intersperse([5, 6], 148)
    # round trip of the code normalisation through the build-defined inverse
    codes = rng.integers(0, 1024, (3, 8, 50))
    assert np.array_equal(oc.denormalise_to_codes(oc.normalise_codes(codes)), codes)


def test_dataset_from_tar_and_loader(tmp_path):
    import io
    import tarfile
    from prompt_tts_amd.tts.dataloader import SingleSpeakerDataset, create_dataloader
    rng = np.random.default_rng(1)
    path = tmp_path / "d.tar"
    with tarfile.open(path, "w") as tf:
        def add(name, data):
            ti = tarfile.TarInfo(name); ti.size = len(data); tf.addfile(ti, io.BytesIO(data))
        for u in ("a", "b", "c"):
            buf = io.BytesIO(); np.save(buf, rng.integers(0, 1024, (2, 20))); add(f"{u}.npy", buf.getvalue())
            add(f"{u}.txt", b"hello world"); add(f"{u}.len.txt", b"19.0")
            if u != "c":
                add(f"{u}.normalized.txt", b"hello world")
    os.environ.pop("PT_CMUDICT", None)
    with pytest.raises(FileNotFoundError):                          # no CMU dictionary file reachable from here: says so
        SingleSpeakerDataset(str(path))
    dl = create_dataloader(str(path), 2, 8, text_to_ids=lambda s: [ord(c) % 147 + 1 for c in s])
    b = next(iter(dl))
    assert b["code"].shape == (2, 2, 20) and b["cmu_sequence_id"].shape == (2, 8)
    assert b["cmu_sequence"][0][0] == 148 and float(b["code"].abs().max()) <= 1.0


def _ljs_like_tar(path, n=7, with_cmu=True):
    """A tar laid out like generate_code.py / encode_codec.py write it: codes + lengths first, every text at the END."""
    import io
    import tarfile
    rng = np.random.default_rng(5)
    with tarfile.open(path, "w") as tf:
        def add(name, data):
            ti = tarfile.TarInfo(name); ti.size = len(data); tf.addfile(ti, io.BytesIO(data))
        for i in range(n):
            buf = io.BytesIO(); np.save(buf, rng.integers(0, 1024, (4, 30 + i))); add(f"LJ{i:03d}.npy", buf.getvalue())
            add(f"LJ{i:03d}.len.txt", str(float(30 + i)).encode())
            if with_cmu:
                buf = io.BytesIO(); np.save(buf, rng.integers(1, 148, 5 + i)); add(f"LJ{i:03d}.cmu.npy", buf.getvalue())
        for i in range(n):
            add(f"LJ{i:03d}.txt", f"utterance number {i}".encode())
            add(f"LJ{i:03d}.normalized.txt", f"utterance number {i} normalised".encode())     # all or none, as the reference needs


def test_lazy_tar_dataset_equals_in_ram_dataset(tmp_path):
    """the offset-indexed dataset yields exactly the items of the whole-tar-in-RAM one, also through worker
    processes and a shuffled loader."""
    from prompt_tts_amd.tts.dataloader import LazySingleSpeakerDataset, SingleSpeakerDataset, create_dataloader
    path = str(tmp_path / "ljs.tar")
    _ljs_like_tar(path)
    ram, lazy = SingleSpeakerDataset(path), LazySingleSpeakerDataset(path)
    assert len(ram) == len(lazy) == 7
    for i in (3, 0, 6, 1, 5, 2, 4):                                # random access order
        a, b = ram[i], lazy[i]
        assert a.keys() == b.keys() and a["text"] == b["text"] and a["cmu_sequence"] == b["cmu_sequence"]
        assert a["code_length"] == b["code_length"] and np.array_equal(a["code"], b["code"])
        assert a.get("text_norm") == b.get("text_norm")
    os.environ.pop("PT_CMUDICT", None)
    with pytest.raises(FileNotFoundError):
        bare = str(tmp_path / "bare.tar"); _ljs_like_tar(bare, with_cmu=False); LazySingleSpeakerDataset(bare)

    class FixedT(torch.utils.data.Dataset):                          # codes of one length, as the 12 s windows of the reference
        def __init__(self, ds): self.ds = ds
        def __len__(self): return len(self.ds)
        def __getitem__(self, i):
            it = dict(self.ds[i]); it["code"] = it["code"][:, :30]; return it
    ref = [b for b in create_dataloader(None, 3, 16, dataset=FixedT(ram))]
    for workers in (0, 2):
        got = [b for b in create_dataloader(None, 3, 16, dataset=FixedT(lazy), num_workers=workers)]
        assert len(got) == len(ref) == 3
        for x, y in zip(ref, got):
            assert torch.equal(x["code"], y["code"]) and torch.equal(x["cmu_sequence_id"], y["cmu_sequence_id"])
            assert torch.equal(x["attention_mask"], y["attention_mask"]) and x["text"] == y["text"]
    assert len(create_dataloader(path, 2, 16, lazy=True).dataset) == 7


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "prompt_tts_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(pt_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 27
    from prompt_tts_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared - {"pt_abi_version", "pt_status_string", "pt_last_hip_error", "pt_struct_size", "pt_wgrad_group_ws_floats"} == set(_lib.SIGNATURES)   # binding == header
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pt_\w+)", out))
    assert exported >= declared
    assert _lib.lib.pt_abi_version() >= 4
    assert b"aligned" in _lib.lib.pt_status_string(-4)
    # argument validation returns before any HIP call: safe without a GPU
    assert _lib.lib.pt_gemm(None, 1, None) == -5
    d = _lib.pt_gemm_desc(); d.M = d.N = d.K = 0
    assert _lib.lib.pt_gemm(ctypes.byref(d), 1, None) == -1
    assert _lib.lib.pt_gemm(ctypes.byref(d), 7, None) == -2
    assert _lib.lib.pt_attn_fwd(None, 1, None) == -5
    assert _lib.lib.pt_wgrad_group(None, 1, None, 0, 0, None) == -5 and _lib.lib.pt_wgrad_group_ws_floats(0) == 256 * 65536
    for i, st in enumerate((_lib.pt_operand, _lib.pt_gemm_desc, _lib.pt_attn_desc, _lib.pt_param_seg, _lib.pt_rowconv_desc,
                            _lib.pt_lstm2_desc, _lib.pt_fold_seg, _lib.pt_encodec_tail_desc, _lib.pt_encodec_stage_desc, _lib.pt_transpose_seg)):
        assert _lib.lib.pt_struct_size(i) == ctypes.sizeof(st)          # the ctypes mirror matches the C layout


def test_cabi_refuses_bad_arguments_with_a_status_never_a_launch():
    """Every entry point validates before it touches HIP: misuse is a negative status, not an exception or a fault."""
    from prompt_tts_amd import _lib
    lib, L = _lib.lib, _lib
    OKP = 0x10000                                            # a 16-byte aligned non-null "pointer" that is never dereferenced
    d = L.pt_gemm_desc(); d.M, d.N, d.K = 128, 128, 64
    d.A.p, d.A.ld, d.B.p, d.B.ld, d.C, d.ldc, d.split_k, d.alpha = OKP, 64, OKP, 64, OKP, 128, 1, 1.0
    bad = L.pt_gemm_desc.from_buffer_copy(d); bad.A.p = OKP + 2
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16, None) == -4                    # misaligned operand
    bad = L.pt_gemm_desc.from_buffer_copy(d); bad.split_k = 4
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16, None) == -5                    # split-K needs the atomic output kind
    bad = L.pt_gemm_desc.from_buffer_copy(d); bad.arow_sum = OKP; bad.arow_n = 64
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16, None) == -5                    # fused bias gradient only on the wgrad
    bad = L.pt_gemm_desc.from_buffer_copy(d); bad.A.kind = L.PT_V_CONV; bad.A.taps = 3; bad.A.cin = 4
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16, None) == -1                    # conv channels must be whole 16-byte chunks
    bad = L.pt_gemm_desc.from_buffer_copy(d); bad.A.kind = L.PT_V_CONV; bad.A.taps = 4; bad.A.cin = 64; bad.A.n_out = 8; bad.A.n_in = 16
    bad.A.rowmap = L.PT_MAP_STRIDED_REFLECT; bad.A.stride = 0
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16, None) == -1                    # strided map without a stride
    assert lib.pt_layernorm_fwd(OKP, OKP, OKP, OKP, OKP, OKP, 16, 12, 1e-5, L.PT_BF16, None) == -1       # C % 8
    assert lib.pt_layernorm_bwd(OKP, OKP, OKP, OKP, OKP, None, OKP, OKP, OKP, 16, 64, 0, 0, L.PT_BF16, None) == -5   # n_rep < 1
    assert lib.pt_colsum(OKP, 64, OKP, 0, 16, 64, 16, 4, 0, L.PT_BF16, None) == -5                        # replicas need a stride
    assert lib.pt_groupnorm_fwd(OKP, None, OKP, OKP, OKP, OKP, OKP, 2, 16, 64, 0, 32, -1.0, 1, L.PT_BF16, None) == -1   # eps < 0
    assert lib.pt_rvq_search(OKP, OKP, OKP, OKP, 2, 8, 10, 8, 1024, 128, None) == -1                      # stage index out of range
    assert lib.pt_ddpm_step(None, OKP, None, OKP, 10, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, None) == -5
    assert lib.pt_fold_replicas(OKP, OKP, None, 3, 16, 64, None) == -5
    r = L.pt_rowconv_desc(); r.B, r.n_rows, r.N, r.cin, r.taps, r.rowmap = 2, 16, 96, 32, 3, L.PT_MAP_CAUSAL_REFLECT
    assert lib.pt_rowconv(ctypes.byref(r), L.PT_BF16, None) == -1                   # more than 64 output channels
    l = L.pt_lstm2_desc(); l.B, l.T, l.H = 4, 10, 300
    assert lib.pt_lstm2_forward(ctypes.byref(l), L.PT_BF16, None) == -1             # hidden size must be a multiple of 256
    assert lib.pt_geglu_fwd(OKP, None, OKP, 16, 12, 0, L.PT_BF16, None) == -1
    assert lib.pt_geglu_fwd(OKP, None, OKP, 16, 16, 0, 9, None) == -2                # unknown dtype
    assert lib.pt_geglu_fwd(OKP, None, OKP, 16, 16, 1, L.PT_BF16, None) == -1       # interleaved needs F % 32 == 0
    # round 4: split storage (PT_BF16X2) and the folded decode step
    x2 = L.pt_gemm_desc.from_buffer_copy(d); x2.M, x2.N, x2.K, x2.ldc, x2.A.ld, x2.B.ld = 128, 128, 64, 256, 128, 128
    bad = L.pt_gemm_desc.from_buffer_copy(x2); bad.A.trans = 1
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16X2, None) == -5                  # plane operands: forward GEMMs only
    bad = L.pt_gemm_desc.from_buffer_copy(x2); bad.ldc = 128
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16X2, None) == -5                  # a plane row holds 2 N elements
    bad = L.pt_gemm_desc.from_buffer_copy(x2); bad.x2_block = 48
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16X2, None) == -5                  # the plane block must divide N
    bad = L.pt_gemm_desc.from_buffer_copy(x2); bad.K = 72
    assert lib.pt_gemm(ctypes.byref(bad), L.PT_BF16X2, None) == -5                  # a k-tile carries both planes of 32 columns
    assert lib.pt_gemm(ctypes.byref(x2), 7, None) == -2                             # unknown dtype
    l2 = L.pt_lstm2_desc(); l2.B, l2.T, l2.H, l2.per_step = 4, 10, 512, 1
    for f in ("x", "xg0", "whh0", "wcat1", "bias1", "h0_seq", "h1_seq", "c0", "c1", "out_elu"):
        setattr(l2, f, OKP)
    assert lib.pt_lstm2_forward(ctypes.byref(l2), L.PT_BF16X2, None) == -1          # plane rows: the persistent f32-class form only
    dl = L.pt_decode_linear_desc(); dl.M, dl.N, dl.K, dl.x, dl.ldx, dl.w, dl.ldw, dl.y, dl.ldy = 64, 512, 512, OKP, 512, OKP, 512, OKP, 512
    bad = L.pt_decode_linear_desc.from_buffer_copy(dl); bad.M = 65
    assert lib.pt_decode_linear(ctypes.byref(bad), None) == -1                      # at most 64 rows
    bad = L.pt_decode_linear_desc.from_buffer_copy(dl); bad.K = 384
    assert lib.pt_decode_linear(ctypes.byref(bad), None) == -1                      # K walks in chunks of 256
    bad = L.pt_decode_linear_desc.from_buffer_copy(dl); bad.ln_gamma = OKP; bad.ln_beta = OKP; bad.K = 1024
    assert lib.pt_decode_linear(ctypes.byref(bad), None) == -1                      # LayerNorm prologue: K = 256 or 512
    bad = L.pt_decode_linear_desc.from_buffer_copy(dl); bad.seg_cols = 512; bad.N = 1536
    assert lib.pt_decode_linear(ctypes.byref(bad), None) == -5                      # column segments need their destinations
    assert lib.pt_run_ops(None, 3, None) == -5
    op = L.pt_op(); op.kind = 99; op.desc = OKP
    assert lib.pt_run_ops(ctypes.byref(op), 1, None) == -5                          # unknown op kind
    assert lib.pt_ar_advance(OKP, OKP, OKP, OKP, OKP, 0, 8, 16, None) == -1
    assert lib.pt_row_select(OKP, 64, None, OKP, 64, None) == -5


def test_product_never_imports_oracle_or_reference():
    pat = re.compile(r"^\s*(from|import)\s+(oracle|tts\.)", re.M)
    for base in ("prompt_tts_amd",):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h")):
                    txt = open(os.path.join(dp, f)).read()
                    assert not pat.search(txt), f"{f} imports the oracle"
                    assert "/root/reference" not in txt
    assert "oracle" not in open(os.path.join(ROOT, "train.py")).read()


def test_model_constructs_on_cpu_with_reference_keys_and_refuses_cpu_compute():
    from oracle import model as om
    from prompt_tts_amd.tts.models import TTSSingleSpeaker
    cfg = om.make_config(d=64, L=1, text_layers=1, n_q=2, T=64, S=32)
    ref = om.TTSSingleSpeaker(cfg)
    m = TTSSingleSpeaker(cfg)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(ref.state_dict())                      # checkpoints interchange with the reference layout
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 64), 3, torch.zeros(1, 32, dtype=torch.int32), torch.ones(1, 32, dtype=torch.int32))
    with pytest.raises(ValueError):
        TTSSingleSpeaker(dict(cfg, up_block_types=["UpBlock1D", "Bogus"]))
    with pytest.raises(ValueError):
        TTSSingleSpeaker(dict(cfg, attention_head_dim=48))


def test_train_lr_lambda_matches_oracle():
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_mod", os.path.join(ROOT, "train.py"))
    src = open(os.path.join(ROOT, "train.py")).read()
    ns = {}
    exec(compile(src[src.index("def lr_lambda"):src.index("def main")], "train_lr", "exec"), {"math": __import__("math")}, ns)
    from oracle.blocks import lr_lambda as want
    for name in ("constant", "constant_with_warmup", "linear", "cosine"):
        f, g = ns["lr_lambda"](name, 5, 40), want(name, 5, 40)
        assert all(abs(f(s) - g(s)) < 1e-12 for s in range(0, 45))


@pytest.mark.parametrize("n_items,bs,world", [(23, 4, 2), (23, 4, 3), (16, 4, 2), (5, 4, 4), (1, 8, 2), (33, 8, 8)])
def test_sharded_batch_sampler_equal_full_batches(n_items, bs, world):
    """accelerate's policy for the reference's loader (train.py:67-69): rank r takes batches r, r+W, ...; the tail wraps
    around to the start of the epoch's order so every rank runs the same number of full-size steps (an odd batch count used
    to leave the low ranks alone inside a collective at the end of an epoch)."""
    from prompt_tts_amd.tts.dataloader import ShardedBatchSampler
    order = list(np.random.default_rng(3).permutation(n_items))
    shards = [list(ShardedBatchSampler(order, bs, r, world)) for r in range(world)]
    n_batches = -(-n_items // bs)
    per_rank = -(-n_batches // world)
    assert all(len(s) == per_rank == len(ShardedBatchSampler(order, bs, r, world)) for r, s in enumerate(shards))
    assert all(len(b) == bs for s in shards for b in s)                       # every step has the full per-rank batch
    # interleaving the ranks' batches gives back the epoch order, then its beginning again (wrap-around)
    flat = [i for k in range(per_rank) for r in range(world) for i in shards[r][k]]
    assert flat[:n_items] == order
    cyc = (order * (len(flat) // n_items + 2))
    assert flat[n_items:] == cyc[:len(flat) - n_items]
    with pytest.raises(ValueError):
        ShardedBatchSampler(order, bs, world, world)


def test_create_dataloader_shards_indices_not_collated_batches():
    from prompt_tts_amd.tts.dataloader import SyntheticDataset, create_dataloader
    ds = SyntheticDataset(10, 2, 16, max_text=32)
    calls = []
    orig = ds.__class__.__getitem__
    class Counting(ds.__class__):
        def __getitem__(self, i):
            calls.append(i); return orig(self, i)
    ds.__class__ = Counting
    dl = create_dataloader(None, 4, 32, shuffle=False, dataset=ds, rank=1, world=2)
    batches = list(dl)
    assert len(batches) == len(dl) == 2 and all(b["code"].shape[0] == 4 for b in batches)
    # 10 items, batches of 4, 2 ranks: [0..3] [4..7] [8,9,0,1] [2,3,4,5]; rank 1 reads ONLY its own two batches
    assert calls == [4, 5, 6, 7, 2, 3, 4, 5]


def test_async_checkpoint_writer_and_adamw_state_dict_format(tmp_path):
    """optim_N.pt has torch.optim.AdamW.state_dict() structure (train.py:142) and round-trips through the flat store;
    the writer snapshots first and serialises in the background."""
    from types import SimpleNamespace
    from prompt_tts_amd import checkpoint
    lin = torch.nn.Linear(4, 3); frozen = torch.nn.Parameter(torch.zeros(5))
    params = [lin.weight, frozen, lin.bias]
    info, off = {}, 0
    for p, fz, name in zip(params, (False, True, False), ("w", "proj_out.bias", "b")):
        info[id(p)] = {"off": off, "n": p.numel(), "frozen": fz, "name": name}; off += 64
    store = SimpleNamespace(info=info, adam_m=torch.arange(off, dtype=torch.float32), adam_v=torch.arange(off, dtype=torch.float32) * 2,
                            step_count=7, device=torch.device("cpu"), flat_p=torch.zeros(off), names=["w", "proj_out.bias", "b"],
                            params_in_model_order=lambda: params)
    hyper = dict(lr=1e-5, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    sd = checkpoint.adamw_state_dict(store, 5e-6, hyper)
    assert sorted(sd["state"]) == [0, 2] and sd["param_groups"][0]["params"] == [0, 1, 2]          # the unused parameter: no state
    assert sd["state"][0]["exp_avg"].shape == (3, 4) and float(sd["state"][2]["step"]) == 7.0
    ref = torch.optim.AdamW(params, **hyper)
    ref.load_state_dict(sd)                                                                        # torch accepts the structure
    assert set(ref.state_dict()["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "params"}
    w = checkpoint.AsyncCheckpointWriter()
    path = str(tmp_path / "optim_1.pt")
    w.save(path, sd)
    store.adam_m.zero_()                                                                           # after save(): not in the file
    w.wait()
    back = torch.load(path)
    assert torch.equal(back["state"][0]["exp_avg"], torch.arange(12, dtype=torch.float32).view(3, 4)) and not os.path.exists(path + ".tmp")
    store.adam_v.zero_(); store.step_count = 0
    checkpoint.load_adamw_state_dict(store, back)
    assert store.step_count == 7 and torch.equal(store.adam_m[:12], torch.arange(12, dtype=torch.float32))
    assert torch.equal(store.adam_v[128:131], torch.arange(128, 131, dtype=torch.float32) * 2) and float(store.adam_m[64:69].abs().sum()) == 0
    w.save(str(tmp_path / "no_such_dir" / "x.pt"), {"a": 1})
    with pytest.raises(Exception):
        w.wait()


def test_flat_format_optimizer_checkpoint_permutes_conv_moments():
    """the flat Adam moments are indexed like the gradient (Conv1d k=3 weights tap-major [Cout][3][Cin]); a round-1
    flat-format file holds them in master order (Cout, Cin, 3).  Loading one must land every conv moment at its gradient-order
    place (what the torch-format path does through _moment_view), not copy the buffer verbatim."""
    from types import SimpleNamespace
    from prompt_tts_amd import checkpoint
    conv = torch.nn.Conv1d(4, 2, 3); lin = torch.nn.Linear(4, 3)
    params = [conv.weight, conv.bias, lin.weight]
    info, off = {}, 0
    for p, name in zip(params, ("c.w", "c.b", "l.w")):
        info[id(p)] = {"off": off, "n": p.numel(), "frozen": False, "name": name}; off += 64
    store = SimpleNamespace(info=info, adam_m=None, adam_v=None, step_count=0, device=torch.device("cpu"), flat_p=torch.zeros(off),
                            names=["c.w", "c.b", "l.w"], params_in_model_order=lambda: params)
    m_master = torch.arange(off, dtype=torch.float32)
    checkpoint.load_adamw_state_dict(store, {"names": ["c.w", "c.b", "l.w"], "exp_avg": m_master, "exp_avg_sq": m_master * 3, "step": 5})
    assert store.step_count == 5
    want = m_master[:24].view(2, 4, 3)                                       # the file: (Cout, Cin, 3)
    got = store.adam_m[:24].view(2, 3, 4)                                    # the store: [Cout][3][Cin]
    assert torch.equal(got, want.permute(0, 2, 1)) and not torch.equal(store.adam_m[:24], m_master[:24])
    assert torch.equal(store.adam_m[64:66], m_master[64:66]) and torch.equal(store.adam_v[128:140], 3 * m_master[128:140])
    # and it agrees with the torch-format path
    sd = checkpoint.adamw_state_dict(store, 1e-5, dict(lr=1e-5, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8))
    assert torch.equal(sd["state"][0]["exp_avg"], want)
    with pytest.raises(RuntimeError):
        checkpoint.load_adamw_state_dict(store, {"names": ["c.w", "c.b", "l.w"], "exp_avg": m_master[:10], "exp_avg_sq": m_master[:10], "step": 5})


def test_wgrad_queue_deferred_problems_never_share_a_launch_with_their_partner(monkeypatch):
    """engine._WgradQueue: a deferred problem (the boundary corrections of a flat conv weight gradient accumulate into the same
    destination as their conv; one launch's fold adds partials with plain read-modify-writes) joins the NEXT group of its class,
    and the final flush drains everything."""
    import torch
    from prompt_tts_amd import engine as E
    launches, ordered_calls = [], []
    monkeypatch.setattr(E.ops, "wgrad_group", lambda descs, ws, wgs: launches.append(list(descs)))
    monkeypatch.setattr(E, "on_side_stream", lambda fn, *tensors, ordered=False: (ordered_calls.append(ordered), fn())[1])
    monkeypatch.setattr(E.ops, "wgrad_group_ws_floats", lambda wgs: 16)
    q = E._WgradQueue(torch.device("cpu"))
    for conv in range(7):                                 # 7 convs: main problem + two deferred corrections each
        q.add(1, f"main{conv}", 12, ())
        q.add(1, f"c{conv}a", 4, (), defer=True)
        q.add(1, f"c{conv}b", 4, (), defer=True)
    q.add(0, "plain", 8, ())
    q.flush()
    flat = [d for grp in launches for d in grp]
    assert sorted(flat) == sorted([f"main{i}" for i in range(7)] + [f"c{i}{s}" for i in range(7) for s in "ab"] + ["plain"])
    assert all(len(grp) <= E.ops.WGRAD_GROUP_MAX for grp in launches)
    where = {d: gi for gi, grp in enumerate(launches) for d in grp}
    for i in range(7):
        assert where[f"c{i}a"] > where[f"main{i}"] and where[f"c{i}b"] > where[f"main{i}"]     # strictly later launches
    assert not q.lists[0] and not q.lists[1] and not q.deferred[0] and not q.deferred[1]
    # every group asks for the ONE ordered side stream (shared slab workspace, read-modify-write folds): PT_SIDE_STREAMS > 1
    # must not let two groups overlap
    assert ordered_calls and all(ordered_calls)


def test_importing_the_package_first_loads_one_hip_runtime():
    """The PyTorch-ROCm wheel bundles its own libamdhip64.so.7; the C-ABI library resolves the same soname.  Imported before
    torch it used to pull in /opt/rocm's copy as a SECOND HIP runtime (every launch then failed with "no ROCm-capable device");
    prompt_tts_amd._lib therefore imports torch before it loads the library."""
    import subprocess
    import sys
    code = ("import prompt_tts_amd, torch\n"
            "maps = open('/proc/self/maps').read().split('\\n')\n"
            "print(sorted({l.split()[-1] for l in maps if 'libamdhip64' in l}))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    libs = eval(out.stdout.strip().split("\n")[-1])
    assert len(libs) == 1, libs
