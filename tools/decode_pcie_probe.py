"""PCIe-inclusive decode rate of BASELINE configs[3] (pinned host codes in, pinned host waveform out; GPU box)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decode_codec import random_decoder_weights
from prompt_tts_amd.encodec import EncodecDecoder
dev = torch.device("cuda:0")
dec = EncodecDecoder(random_decoder_weights(0), device=dev, dtype=torch.bfloat16)
codes_h = torch.randint(0, 1024, (64, 8, 1024), generator=torch.Generator().manual_seed(7)).pin_memory()
wav_h = torch.empty(64, 1, 327680, dtype=torch.float32).pin_memory()
for _ in range(3):
    wav_h.copy_(dec.decode(codes_h.to(dev, non_blocking=True)), non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    wav_h.copy_(dec.decode(codes_h.to(dev, non_blocking=True)), non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"decode 64 x 1024 incl. pinned H2D codes (4 MB) + D2H waveform (84 MB): {dt * 1e3:.2f} ms = {64 * 1024 / 75 / dt / 1e3:.1f} k audio-s/s")
