#!/usr/bin/env python
"""Codec codes -> waveform, with the reference's surface (decode_codec.py:12-33): `decode(encoded_frames)` and
`python decode_codec.py --npy_path x.npy` writing x.wav at 24 kHz.  The decoder runs on the MI355X kernels
(prompt_tts_amd/encodec.py).

The reference builds `EncodecModel.encodec_model_24khz()` at import time, which downloads pretrained weights;
this build never fetches anything: pass `--weights <encodec state_dict .pt>` (original `encodec` package naming,
weight_g/weight_v are folded on load).  A seeded RANDOM decoder (noise out; plumbing and benchmarks only) is used only when
asked for explicitly: `--random_weights` / `load_decoder(None, random_weights=True)`.
"""
from argparse import ArgumentParser

import numpy as np
import torch

from prompt_tts_amd.encodec import EncodecDecoder, weights_from_encodec_state_dict

_model = None


def load_decoder(weights_path=None, dtype=torch.float32, device="cuda", seed=0, random_weights=False):
    """dtype: torch.float32 (default) = the reference's precision (decode_codec.py:12-16 decodes in fp32): f32-class arithmetic,
    1e-3-exact against it; torch.bfloat16 = 1.7x faster at ~1e-2 of the waveform peak (a preview mode)."""
    global _model
    if weights_path is not None:
        W = weights_from_encodec_state_dict(torch.load(weights_path, map_location="cpu"))
    elif random_weights:
        W = random_decoder_weights(seed)
    else:
        raise RuntimeError("no Encodec weights: pass --weights / load_decoder(<encodec state_dict .pt>) -- the pretrained 24 kHz "
                           "model the reference downloads is never fetched here (random_weights=True gives a seeded random decoder)")
    _model = EncodecDecoder(W, device=device, dtype=dtype)
    return _model


def random_decoder_weights(seed=0, n_q=8):
    """Seeded fan-in-scaled weights of the 24 kHz architecture (no checkpoint is available offline)."""
    g = torch.Generator().manual_seed(seed)

    def t(*shape, fan=None):
        fan = fan or (shape[1] * (shape[2] if len(shape) > 2 else 1))
        return (torch.rand(shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5

    W = {"codebooks": torch.randn(n_q, 1024, 128, generator=g) * 0.5, "conv0.w": t(512, 128, 7), "conv0.b": t(512, fan=100)}
    for l in range(2):
        W[f"lstm.w_ih{l}"] = t(2048, 512); W[f"lstm.w_hh{l}"] = t(2048, 512)
        W[f"lstm.b_ih{l}"] = t(2048, fan=100); W[f"lstm.b_hh{l}"] = t(2048, fan=100)
    C = 512
    for i, r in enumerate((8, 5, 4, 2)):
        W[f"up{i}.w"] = t(C, C // 2, 2 * r, fan=2 * C); W[f"up{i}.b"] = t(C // 2, fan=100)
        C //= 2
        W[f"res{i}.c3.w"] = t(C // 2, C, 3); W[f"res{i}.c3.b"] = t(C // 2, fan=100)
        W[f"res{i}.c1.w"] = t(C, C // 2, 1); W[f"res{i}.c1.b"] = t(C, fan=100)
        W[f"res{i}.sc.w"] = t(C, C, 1); W[f"res{i}.sc.b"] = t(C, fan=100)
    W["final.w"] = t(1, 32, 7); W["final.b"] = t(1, fan=100)
    return W


def decode(encoded_frames: torch.Tensor):
    if len(encoded_frames.shape) != 3:
        raise BaseException("The encoded_frames must have the shape of [B, N_q, T]")
    if _model is None:
        raise RuntimeError("decode(): no decoder loaded -- call load_decoder(<encodec state_dict .pt>) first (the reference downloads "
                           "pretrained weights at import time; this build never fetches anything)")
    return _model.decode(encoded_frames)


def write_wav(path, wav, sample_rate=EncodecDecoder.sample_rate):
    """(1, n) or (n,) float waveform in [-1, 1] -> 16-bit PCM file (what soundfile writes by default in the reference)."""
    from scipy.io import wavfile
    mono = wav.reshape(-1).clamp(-1, 1).cpu().numpy()
    wavfile.write(path, sample_rate, (mono * 32767.0).astype(np.int16))


def run_cli(npy_path, weights=None, dtype="f32", random_weights=False):
    """`--npy_path x.npy` -> x.wav (first item of the batch), as the reference's command line does."""
    codes = torch.from_numpy(np.load(npy_path))
    if codes.dim() == 2:                          # a single utterance saved as [N_q, T]
        codes = codes[None]
    load_decoder(weights, {"f32": torch.float32, "bf16": torch.bfloat16}[dtype], random_weights=random_weights)
    wav = decode(codes)
    out_path = npy_path[:-4] + ".wav" if npy_path.endswith(".npy") else npy_path + ".wav"
    write_wav(out_path, wav[0])
    return out_path


if __name__ == "__main__":
    cli = ArgumentParser(description="Codec codes (.npy, [N_q, T] or [B, N_q, T]) -> 24 kHz waveform next to the input file.")
    cli.add_argument("--npy_path", required=True, help="codec code matrix written by the data preparation")
    cli.add_argument("--weights", default=None, help="encodec state_dict (.pt) in the original package's naming")
    cli.add_argument("--random_weights", action="store_true", help="seeded random decoder (writes noise: plumbing tests only)")
    cli.add_argument("--dtype", default="f32", choices=["bf16", "f32"],
                     help="f32 (default): the reference's fp32 precision (f32-class arithmetic); bf16: faster, ~1e-2 of the waveform peak")
    ns = cli.parse_args()
    if ns.weights is None and not ns.random_weights:
        cli.error("--weights <encodec state_dict .pt> is required (or --random_weights for a plumbing run)")
    print(run_cli(ns.npy_path, ns.weights, ns.dtype, ns.random_weights))
