"""One command, N ranks: what `accelerate launch train.py` does for the reference (README.md:36-42, train.py:25-29).

`spawn_ranks(n, script, argv)` starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node n script argv...` as a CHILD
(rendezvous on 127.0.0.1, a free port) and returns its exit code.  It must be called before the calling process has made any
torch.cuda / HIP call: the ranks are fresh processes, the caller never touches the GPU and never exec()s.
"""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, script, argv):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def local_device_index(backend):
    """LOCAL_RANK -> device index.  RCCL needs one device per rank (two ranks on one device fail inside the first collective,
    not with a message); the gloo rehearsal backends may share devices."""
    import torch
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if backend == "nccl":
        if local >= max(ndev, 1):
            raise SystemExit(f"LOCAL_RANK={local} but only {ndev} GPU(s) are visible: RCCL needs one device per rank "
                             "(rehearse more ranks than GPUs with the gloo backend)")
        return local
    return local % max(ndev, 1)
