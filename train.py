#!/usr/bin/env python
"""Training entry point with the reference's CLI and 1d_config schema (train.py:21-172), MI355X-native inside.

  accelerate launch train.py --data_file d.tar --log_dir logs --config_file 1d_config.json --ckpt_dir ckpt/
  python -m torch.distributed.run --nproc-per-node 8 train.py ...        (same thing; one process per GPU)
  python train.py --gpus 8 ...                                           (same thing; starts its own ranks)
  python train.py --synthetic 256 --config_file ... --log_dir ... --ckpt_dir ...   (no tar needed)

Differences from the reference, all behind the same surface: the step is the fused HIP path
(TTSSingleSpeaker.train_step: add_noise -> forward -> MSE -> backward -> clip -> AdamW, no host sync except the
logged loss), data parallelism is a bucketed RCCL all-reduce of the flat grad buffer overlapped with backward
(prompt_tts_amd/parallel.py) instead of DDP(find_unused_parameters=True), and the batch sharding follows
accelerate's round-robin over batches (rank r takes batches r, r+W, ...; the tail wraps around so that every rank
runs the same number of steps) at the INDEX level: a rank collates only its own batches.
"""
import argparse
import json
import logging
import math
import os

import torch

from prompt_tts_amd import checkpoint, engine, launch, parallel
from prompt_tts_amd.tts.dataloader import DeviceFeeder, SyntheticDataset, create_dataloader
from prompt_tts_amd.tts.models import TTSSingleSpeaker

logging.basicConfig(format="%(asctime)s - %(levelname)s: %(message)s", level=logging.INFO, datefmt="%I:%M:%S")

ADAMW = dict(lr=1e-5, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)      # hard-coded in the reference (train.py:41-47)


def lr_lambda(name, num_warmup_steps, num_training_steps):
    """diffusers.optimization.get_scheduler multipliers for the names the 1d_config may carry (train.py:60-65)."""
    w, n = num_warmup_steps, num_training_steps
    if name == "constant":
        return lambda s: 1.0
    if name == "constant_with_warmup":
        return lambda s: float(s) / float(max(1.0, w)) if s < w else 1.0
    if name == "linear":
        return lambda s: float(s) / float(max(1, w)) if s < w else max(0.0, float(n - s) / float(max(1, n - w)))
    if name == "cosine":
        return lambda s: (float(s) / float(max(1, w)) if s < w else
                          max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(s - w) / float(max(1, n - w))))))
    raise ValueError(f"unsupported lr_scheduler {name!r}")


def main(args):
    config = json.load(open(args.config_file, "r"))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("PT_TRAIN_BACKEND", "nccl")                 # "gloo": rehearse the N-rank path on one GPU
    local = launch.local_device_index(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            torch.distributed.init_process_group(backend)
    try:
        from torch.utils.tensorboard import SummaryWriter
        writer = SummaryWriter(log_dir=args.log_dir) if rank == 0 else None
    except Exception:                                                       # tensorboard is optional here
        writer = None

    torch.manual_seed(0)
    model = TTSSingleSpeaker(config, dtype=torch.float32 if args.dtype == "f32" else torch.bfloat16).to(dev)
    reducer = parallel.attach(model) if world > 1 else None           # PT_DP_SHARDED=1: reduce-scatter + sharded optimizer
    sharded = reducer is not None and hasattr(reducer, "owned_ranges")
    ds = SyntheticDataset(args.synthetic, config["in_channels"], config["sample_size"], args.max_seq_length) if args.synthetic else None
    dataloader = create_dataloader(args.data_file, args.batch_size, args.max_seq_length, shuffle=True, dataset=ds,
                                   lazy=args.lazy_tar, num_workers=args.num_workers, rank=rank, world=world)
    accum = config["gradient_accumulation_steps"]
    steps_per_epoch = math.ceil(len(dataloader) / accum)             # len(dataloader) = this rank's batches (equal on all ranks)
    max_train_steps = config["num_train_epochs"] * steps_per_epoch
    lam = lr_lambda(config["lr_scheduler"], config["lr_warmup_steps"] * accum, max_train_steps * accum)
    st = model.store
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    global_step = opt_step = 0
    start_epoch = 0
    if args.resume_epoch:
        # the resume the reference never wrote (it only saves, train.py:139-144): model, Adam moments + step, LR-schedule
        # position and the noise / timestep generator of this rank, from the files written at the end of that epoch
        model.load_state_dict(torch.load(args.ckpt_dir + f"ckpt_{args.resume_epoch}.pt", map_location=dev))
        checkpoint.load_adamw_state_dict(st, torch.load(args.ckpt_dir + f"optim_{args.resume_epoch}.pt", map_location="cpu"))
        extra_path = args.ckpt_dir + f"resume_{args.resume_epoch}.pt"
        extra = torch.load(extra_path, map_location="cpu") if os.path.exists(extra_path) else {}
        opt_step, global_step = int(extra.get("opt_step", st.step_count)), int(extra.get("global_step", st.step_count * accum))
        if "gen_state" in extra and rank < len(extra["gen_state"]):
            gen.set_state(extra["gen_state"][rank].cpu())
        start_epoch = args.resume_epoch
        logging.info(f"resumed after epoch {start_epoch}: optimizer step {opt_step}")
    writer_ckpt = checkpoint.AsyncCheckpointWriter(dev) if rank == 0 else None
    for epoch in range(start_epoch, config["num_train_epochs"]):
        logging.info(f"Starting epoch {epoch}:")
        torch.manual_seed(epoch)                                            # same shuffle on every rank (accelerate C7)
        micro = 0
        loss_acc = torch.zeros(1, device=dev)
        # this rank's batches only (ShardedBatchSampler), pinned and copied to the GPU ahead of the step loop
        n_batches = len(dataloader)
        for bi, batch in enumerate(DeviceFeeder(dataloader, dev) if args.prefetch else dataloader):
            codes = batch["code"].to(dev)
            ids = batch["cmu_sequence_id"].to(dev); mask = batch["attention_mask"].to(dev)
            noise = torch.randn(codes.shape, device=dev, generator=gen)
            t = torch.randint(0, 1000, (codes.shape[0],), device=dev, generator=gen)
            if micro == 0:
                st.zero_grad()
                if reducer is not None:
                    reducer.begin()
            # the optimizer steps every `accum` micro-batches AND on the last batch of the epoch (accelerate syncs at
            # end_of_dataloader), so leftover micro-batches are never dropped
            sync = micro == accum - 1 or bi == n_batches - 1
            model.grad_ready_hook = reducer.on_ready if (reducer is not None and sync) else None
            model.loss_and_backward(codes, noise, t, ids, mask, loss_out=loss_acc,
                                    grad_scale=1.0 / (accum * world))
            micro += 1
            global_step += 1
            if sync:
                if reducer is not None:
                    reducer.finish()
                if sharded:      # reduce-scatter done: AdamW on this rank's slices, all-gather of the new values
                    engine.adamw_step_sharded(st, reducer, ADAMW["lr"] * lam(opt_step), ADAMW["betas"], ADAMW["eps"], ADAMW["weight_decay"], 1.0)
                else:
                    st.adamw_step(ADAMW["lr"] * lam(opt_step), ADAMW["betas"], ADAMW["eps"], ADAMW["weight_decay"], 1.0)
                opt_step += 1; micro = 0
                if opt_step % args.log_every == 0:
                    if world > 1:
                        torch.distributed.all_reduce(loss_acc, op=torch.distributed.ReduceOp.AVG)
                    train_loss = float(loss_acc) / accum                    # the only host sync
                    if rank == 0:
                        logging.info(f"step {opt_step}: MSE={train_loss:.6f}")
                        if writer is not None:
                            writer.add_scalar("Loss/train", train_loss, global_step)
                loss_acc.zero_()
        gen_states = [gen.get_state()]
        if world > 1:
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, gen.get_state().cpu())
            gen_states = gathered
            torch.distributed.barrier()
        if sharded and epoch % config["save_per_epochs"] == 0 and st.adam_m is not None:
            reducer.allgather_published(st.adam_m); reducer.allgather_published(st.adam_v)   # the moments live on their owners
        if rank == 0 and epoch % config["save_per_epochs"] == 0:
            # same file names as the reference, which concatenates ckpt_dir and the name without a separator; optim_N.pt is a
            # torch.optim.AdamW state_dict (train.py:142).  Snapshots go to pinned host memory on a copy stream and are
            # written by a background thread: the next epoch starts at once.
            writer_ckpt.wait()
            writer_ckpt.save(args.ckpt_dir + f"ckpt_{epoch + 1}.pt", model.state_dict())
            writer_ckpt.save(args.ckpt_dir + f"optim_{epoch + 1}.pt", checkpoint.adamw_state_dict(st, ADAMW["lr"] * lam(opt_step), ADAMW))
            writer_ckpt.save(args.ckpt_dir + f"resume_{epoch + 1}.pt",
                             {"opt_step": opt_step, "global_step": global_step, "gen_state": gen_states})
    if writer_ckpt is not None:
        writer_ckpt.wait()
    if writer is not None:
        writer.flush(); writer.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def parse_args():
    p = argparse.ArgumentParser(description="Train TTS models. The data is stored in WebDataset format.")
    p.add_argument("--data_file", type=str, default=None, help="Path to the training data file.")
    p.add_argument("--log_dir", type=str, required=True, help="Directory to save logs.")
    p.add_argument("--config_file", type=str, required=True, help="Path to config file.")
    p.add_argument("--ckpt_dir", type=str, required=True, help="Directory to save checkpoints.")
    p.add_argument("--batch_size", type=int, default=32, help="Per-process batch size.")
    p.add_argument("--max_seq_length", type=int, default=550, help="Maximum length of cmu sequence.")
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic LJSpeech-shaped items instead of a tar")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--log_every", type=int, default=10)
    p.add_argument("--lazy_tar", action="store_true", help="read utterances on demand from an offset index instead of loading the tar")
    p.add_argument("--num_workers", type=int, default=0, help="DataLoader worker processes (collate off the step loop)")
    p.add_argument("--prefetch", type=int, default=1, help="1: pin + copy batches to the GPU two steps ahead (DeviceFeeder)")
    p.add_argument("--resume_epoch", type=int, default=0, help="continue after this epoch from ckpt_dir's ckpt_N.pt / optim_N.pt")
    p.add_argument("--gpus", type=int, default=0,
                   help="N > 1 without a launcher: start N ranks (one per GPU) the way `accelerate launch` does for the reference")
    a = p.parse_args()
    if not a.synthetic and not a.data_file:
        p.error("--data_file is required (or --synthetic N)")
    return a


if __name__ == "__main__":
    _args = parse_args()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # fresh children, started before this process touches the GPU
        raise SystemExit(launch.spawn_ranks(_args.gpus, os.path.abspath(__file__), __import__("sys").argv[1:]))
    main(_args)
