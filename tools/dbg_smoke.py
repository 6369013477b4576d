import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
def peek(tag):
    e = hip.hipPeekAtLastError()
    print(tag, e, hip.hipGetErrorString(e).decode(), flush=True)
from oracle import model as om, train_step as ots
from oracle.init import deterministic_init_
from prompt_tts_amd.tts.models import TTSSingleSpeaker
from prompt_tts_amd import engine as E, ops
peek("start")
assert torch.cuda.is_available()
peek("after is_available")
dev = torch.device("cuda:0")
cfg = om.make_config(d=256, L=1, text_layers=1, n_q=2, T=64, S=32)
g = torch.Generator().manual_seed(0)
B, S = 2, 32
x0 = torch.rand(B, 2, 64, generator=g) * 2 - 1
noise = torch.randn(B, 2, 64, generator=g)
t = torch.randint(0, 1000, (B,), generator=g)
ids = torch.randint(1, 149, (B, S), generator=g, dtype=torch.int32)
mask = torch.ones(B, S, dtype=torch.int32)
ref = deterministic_init_(om.TTSSingleSpeaker(cfg), 1)
lref, gref = ots.train_step(ref, ots.make_optimizer(ref), x0, noise, t, ids, mask)
peek("after oracle step")
m = deterministic_init_(TTSSingleSpeaker(cfg, dtype=torch.float32), 1)
m = m.to(dev)
peek("after to")
orig = ops.pack_shadow
def dbg(*a):
    peek("before pack_shadow")
    return orig(*a)
ops.pack_shadow = dbg
for name in ("zeros", "frombuffer"):
    pass
try:
    st = m.store
    print("store ok")
except Exception as ex:
    print("EXC", ex)
    peek("after exc")
