"""Checkpoint files of the training loop (reference: train.py:139-144) and the resume the reference never wrote.

Per saved epoch N the reference writes, into `ckpt_dir` (path and name concatenated without a separator):
  ckpt_N.pt    model.state_dict()                               -- same keys / shapes here
  optim_N.pt   torch.optim.AdamW(model.parameters()).state_dict() -- written here in exactly that format:
               {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{lr, betas, eps, weight_decay, ..., "params"}]}
               with i = position in model.parameters(); the never-used proj_out parameters have no state entry (their
               .grad is None under the reference too), so torch.optim.AdamW(...).load_state_dict() accepts the file.
  (accelerator.save_state: RNG / scheduler state)  -- here resume_N.pt: optimizer-step counters and each rank's noise generator.

Saving does not stall the step loop: tensors are snapshotted device -> pinned host memory on a copy stream (ordered after the
optimizer step that produced them) and a background thread serialises them; wait() joins before the next save / at exit.
"""
import os
import threading

import torch


def _moment_view(flat, info, p):
    """A parameter's Adam moment in the parameter's own shape.  The flat moments are indexed like the flat GRADIENT buffer: a
    Conv1d k=3 weight's gradient (and so its moments) is stored tap-major [Cout][3][Cin] (include/prompt_tts_hip.h, pt_param_seg)."""
    v = flat[info["off"]:info["off"] + info["n"]]
    if p.dim() == 3 and p.shape[2] == 3:
        return v.view(p.shape[0], 3, p.shape[1]).permute(0, 2, 1)
    return v.view(p.shape)


def adamw_state_dict(store, lr, hyper):
    """torch.optim.AdamW.state_dict() for the flat store: per-parameter views of the flat Adam moments."""
    state, ids = {}, []
    step = torch.tensor(float(store.step_count))
    for i, p in enumerate(store.params_in_model_order()):
        ids.append(i)
        info = store.info[id(p)]
        if info["frozen"] or store.adam_m is None:
            continue
        state[i] = {"step": step.clone(), "exp_avg": _moment_view(store.adam_m, info, p), "exp_avg_sq": _moment_view(store.adam_v, info, p)}
    group = {"lr": lr, "betas": tuple(hyper["betas"]), "eps": hyper["eps"], "weight_decay": hyper["weight_decay"], "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "initial_lr": hyper["lr"], "params": ids}
    return {"state": state, "param_groups": [group]}


def load_adamw_state_dict(store, sd):
    """Inverse of adamw_state_dict (also accepts the flat private format written by round-1 builds)."""
    dev = store.device
    if "exp_avg" in sd and "names" in sd:                                   # round-1 flat format: moments in MASTER order
        if sd["names"] != store.names:
            raise RuntimeError("optimizer checkpoint does not match this model's parameter list")
        # the flat moments are now indexed like the GRADIENT (Conv1d k=3 weights tap-major, [Cout][3][Cin]); a round-1 file holds
        # them in master order (Cout, Cin, 3): go through the per-parameter views, which permute (copying the buffers verbatim
        # would silently scramble the moments of every conv k=3 weight)
        m_old, v_old = sd["exp_avg"].to(dev), sd["exp_avg_sq"].to(dev)
        if m_old.numel() != store.flat_p.numel() or v_old.numel() != store.flat_p.numel():
            raise RuntimeError("optimizer checkpoint (flat format) does not match this model's parameter buffer")
        store.adam_m = torch.zeros_like(store.flat_p); store.adam_v = torch.zeros_like(store.flat_p)
        for p in store.params_in_model_order():
            info = store.info[id(p)]
            if info["frozen"]:
                continue
            lo = info["off"]
            _moment_view(store.adam_m, info, p).copy_(m_old[lo:lo + p.numel()].view(p.shape))
            _moment_view(store.adam_v, info, p).copy_(v_old[lo:lo + p.numel()].view(p.shape))
        store.step_count = int(sd["step"])
        return
    params = store.params_in_model_order()
    ids = sd["param_groups"][0]["params"]
    if len(ids) != len(params):
        raise RuntimeError(f"optimizer checkpoint holds {len(ids)} parameters, the model has {len(params)}")
    store.adam_m = torch.zeros_like(store.flat_p); store.adam_v = torch.zeros_like(store.flat_p)
    steps = set()
    for i, p in zip(ids, params):
        st = sd["state"].get(i)
        if st is None:
            continue
        info = store.info[id(p)]
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise RuntimeError(f"optimizer state {i} has shape {tuple(st['exp_avg'].shape)}, parameter {info['name']} {tuple(p.shape)}")
        _moment_view(store.adam_m, info, p).copy_(st["exp_avg"].to(dev))
        _moment_view(store.adam_v, info, p).copy_(st["exp_avg_sq"].to(dev))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise RuntimeError(f"per-parameter step counts differ ({sorted(steps)}): not a checkpoint of this training loop")
    store.step_count = steps.pop() if steps else 0


class AsyncCheckpointWriter:
    """save(path, obj): snapshot every tensor in the (nested dict / list) `obj` to pinned host memory on a copy stream, then
    torch.save it from a background thread to `path` (written as path + '.tmp', renamed when complete)."""

    def __init__(self, device=None):
        self.device = torch.device(device) if device is not None else None
        self.stream = torch.cuda.Stream(device=self.device) if self.device is not None and self.device.type == "cuda" else None
        self._threads = []
        self.errors = []

    def _snapshot(self, obj):
        if torch.is_tensor(obj):
            if obj.is_cuda:
                host = torch.empty(obj.shape, dtype=obj.dtype, device="cpu", pin_memory=True)
                host.copy_(obj, non_blocking=True)
                return host
            return obj.detach().clone()
        if isinstance(obj, dict):
            out = type(obj)((k, self._snapshot(v)) for k, v in obj.items())
            meta = getattr(obj, "_metadata", None)      # nn.Module.state_dict()'s per-module version table: the saved
            if meta is not None:                        # ckpt_N.pt then loads exactly like the reference's (train.py:141)
                out._metadata = meta
            return out
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._snapshot(v) for v in obj)
        return obj

    def save(self, path, obj):
        done = None
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))     # after the step that produced the tensors
            with torch.cuda.stream(self.stream):
                snap = self._snapshot(obj)
                done = torch.cuda.Event(); done.record(self.stream)
            # the device tensors must stay untouched until the copies ran: the caller's stream waits for the copy stream
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        else:
            snap = self._snapshot(obj)

        def work():
            try:
                if done is not None:
                    done.synchronize()
                tmp = path + ".tmp"
                torch.save(snap, tmp)
                os.replace(tmp, path)
            except BaseException as e:          # surfaced by wait()
                self.errors.append(e)
        th = threading.Thread(target=work, daemon=False)
        th.start()
        self._threads.append(th)

    def wait(self):
        for th in self._threads:
            th.join()
        self._threads = []
        if self.errors:
            e = self.errors[0]; self.errors = []
            raise e
