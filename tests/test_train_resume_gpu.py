"""train.py: reference-compatible checkpoint files, and the resume the reference never wrote."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _config(epochs):
    return {"cmu_vocab_len": 149, "cmu_seq_len": 64, "cross_attention_dim": 256, "attention_head_dim": 64,
            "text_encoder_dropout": 0.0, "text_encoder_layers": 1, "sample_size": 64, "in_channels": 8, "out_channels": 8,
            "layers_per_block": 1, "block_out_channels": [256, 256],      # UNet heads are C/8 wide: 32 is the narrowest
            "down_block_types": ["CrossAttnDownBlock1D", "DownBlock1D"], "mid_block_type": "UNetMidBlock1DCrossAttn",
            "up_block_types": ["UpBlock1D", "CrossAttnUpBlock1D"],
            "gradient_accumulation_steps": 1, "lr_scheduler": "constant_with_warmup", "lr_warmup_steps": 2,
            "num_train_epochs": epochs, "save_per_epochs": 1}


def _run(tmp, name, epochs, resume=0):
    d = os.path.join(tmp, name) + os.sep
    os.makedirs(d, exist_ok=True)
    cfg = os.path.join(d, "cfg.json")
    json.dump(_config(epochs), open(cfg, "w"))
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--synthetic", "12", "--config_file", cfg, "--log_dir", d,
           "--ckpt_dir", d, "--batch_size", "4", "--max_seq_length", "64", "--dtype", "f32"]
    if resume:
        cmd += ["--resume_epoch", str(resume)]
    subprocess.run(cmd, check=True, cwd=ROOT, timeout=600)
    return d


@pytest.mark.gpu
def test_resume_continues_the_same_trajectory(dev, tmp_path):
    straight = _run(str(tmp_path), "straight", 2)
    part = _run(str(tmp_path), "part", 1)
    # same directory: epoch 1's files are there; ask for 2 epochs and resume after the first
    json.dump(_config(2), open(os.path.join(part, "cfg.json"), "w"))
    _run(str(tmp_path), "part", 2, resume=1)
    a = torch.load(os.path.join(straight, "ckpt_2.pt"), map_location="cpu")
    b = torch.load(os.path.join(part, "ckpt_2.pt"), map_location="cpu")
    assert list(a) == list(b) and any(k.startswith("unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q") for k in a)
    # f32 run; only the atomics' summation order differs between processes, which Adam's normalised update turns into +-lr
    # on parameters whose gradient is noise: compare the mean deviation with the mean distance travelled in epoch 2
    first = torch.load(os.path.join(straight, "ckpt_1.pt"), map_location="cpu")
    fl = [k for k in a if a[k].is_floating_point() and "proj_out" not in k and "inv_freq" not in k]
    dev_ = sum(float((a[k].float() - b[k].float()).abs().sum()) for k in fl) / sum(a[k].numel() for k in fl)
    moved = sum(float((a[k].float() - first[k].float()).abs().sum()) for k in fl) / sum(a[k].numel() for k in fl)
    assert moved > 1e-5 and dev_ < 0.05 * moved, (dev_, moved)
    oa = torch.load(os.path.join(straight, "optim_2.pt"), map_location="cpu")
    ob = torch.load(os.path.join(part, "optim_2.pt"), map_location="cpu")
    ra = torch.load(os.path.join(straight, "resume_2.pt"), map_location="cpu")
    rb = torch.load(os.path.join(part, "resume_2.pt"), map_location="cpu")
    assert ra["opt_step"] == rb["opt_step"] == 6
    # optim_N.pt is a torch.optim.AdamW state_dict (train.py:142): a stock optimizer over a same-shaped model loads it
    from prompt_tts_amd.tts.models import TTSSingleSpeaker
    m = TTSSingleSpeaker(_config(2), dtype=torch.float32)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-5, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    opt.load_state_dict(oa)
    named = dict(m.named_parameters()); by_index = list(named)
    assert all("proj_out" not in by_index[i] for i in oa["state"]) and len(oa["param_groups"][0]["params"]) == len(by_index)
    assert all(float(s["step"]) == 6.0 and s["exp_avg"].shape == named[by_index[i]].shape for i, s in oa["state"].items())
    # without the optimizer state the second epoch would restart Adam's moments: make sure they were carried over
    # (restarted moments would be ~half as large after 3 instead of 6 steps; run-to-run noise is a few per cent)
    va = torch.cat([s["exp_avg_sq"].flatten() for s in oa["state"].values()])
    vb = torch.cat([s["exp_avg_sq"].flatten() for s in ob["state"].values()])
    assert float((va - vb).abs().sum()) <= 0.1 * float(va.abs().sum())


@pytest.mark.gpu
def test_device_feeder_delivers_the_same_batches_in_order(dev):
    """The prefetching feeder changes WHERE the copies happen, not what arrives (and surfaces loader errors)."""
    from prompt_tts_amd.tts.dataloader import DeviceFeeder, SyntheticDataset, create_dataloader
    ds = SyntheticDataset(10, 4, 32, 64)
    ref = list(create_dataloader(None, 3, 64, dataset=ds))
    got = list(DeviceFeeder(create_dataloader(None, 3, 64, dataset=ds), dev, depth=2))
    assert len(got) == len(ref) == 4
    for a, b in zip(ref, got):
        assert b["code"].device.type == "cuda" and b["attention_mask"].device.type == "cuda"
        assert torch.equal(a["code"], b["code"].cpu()) and torch.equal(a["cmu_sequence_id"], b["cmu_sequence_id"].cpu())
        assert a["text"] == b["text"] and a["cmu_sequence"] == b["cmu_sequence"]

    class Boom:
        def __iter__(self):
            yield ref[0]
            raise ValueError("broken shard")

        def __len__(self):
            return 2
    it = iter(DeviceFeeder(Boom(), dev))
    next(it)
    with pytest.raises(ValueError, match="broken shard"):
        next(it)


@pytest.mark.gpu
def test_two_ranks_odd_batch_count_end_of_epoch(dev, tmp_path):
    """2 ranks (gloo, sharing the one GPU), 18 items in batches of 4 = 5 batches: an odd count.  Rank r takes batches r, r+2, ...
    and the tail wraps around (accelerate's shard policy for the reference's loader), so both ranks run 3 steps per epoch and
    meet in every collective -- no rank is left alone in an all-reduce at the end of an epoch; gradient accumulation of 2
    leaves one micro-batch over, which is stepped on the last batch instead of being dropped."""
    d = str(tmp_path) + os.sep
    cfg = dict(_config(2)); cfg["gradient_accumulation_steps"] = 2
    json.dump(cfg, open(d + "cfg.json", "w"))
    env = dict(os.environ, PT_TRAIN_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "train.py"), "--synthetic", "18", "--config_file", d + "cfg.json",
           "--log_dir", d, "--ckpt_dir", d, "--batch_size", "4", "--max_seq_length", "64", "--dtype", "f32", "--log_every", "1"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, timeout=600, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    extra = torch.load(d + "resume_2.pt", map_location="cpu")
    # per rank and epoch: 3 micro-batches -> optimizer steps after micro-batch 2 and (end of epoch) after micro-batch 3
    assert extra["global_step"] == 6 and extra["opt_step"] == 4 and len(extra["gen_state"]) == 2
    sd = torch.load(d + "ckpt_2.pt", map_location="cpu")
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [1, 2])
def test_accelerate_launch_entry_point(dev, tmp_path, nproc):
    """The reference is started as `accelerate launch train.py ...` (README.md:36-42, train.py:25-29); north_star keeps that entry
    point.  Run it: accelerate's launcher (1 process: its simple launcher; 2 processes: its torch.distributed.run launcher, both
    ranks on this one GPU over gloo -- PT_TRAIN_BACKEND, as the other two-rank tests) must start train.py, which must train an
    epoch on synthetic items and write the reference's checkpoint files."""
    d = str(tmp_path) + os.sep
    cfg = os.path.join(d, "cfg.json")
    json.dump(_config(1), open(cfg, "w"))
    cmd = [sys.executable, "-m", "accelerate.commands.launch", "--num_processes", str(nproc), "--num_machines", "1",
           "--mixed_precision", "no", "--dynamo_backend", "no"]
    if nproc > 1:
        cmd += ["--multi_gpu", "--main_process_ip", "127.0.0.1", "--main_process_port", "29653"]
    cmd += [os.path.join(ROOT, "train.py"), "--synthetic", "16", "--config_file", cfg, "--log_dir", d, "--ckpt_dir", d,
            "--batch_size", "4", "--max_seq_length", "64", "--dtype", "bf16"]
    env = dict(os.environ, PT_TRAIN_BACKEND="gloo" if nproc > 1 else "nccl", MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, timeout=900, env=env, capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    sd = torch.load(os.path.join(d, "ckpt_1.pt"), map_location="cpu")
    assert "text_encoder.word_embedding.weight" in sd and os.path.exists(os.path.join(d, "optim_1.pt"))
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
