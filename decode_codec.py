#!/usr/bin/env python
"""Codec codes -> waveform, with the reference's surface (decode_codec.py:12-33): `decode(encoded_frames)` and
`python decode_codec.py --npy_path x.npy` writing x.wav at 24 kHz.  The decoder runs on the MI355X kernels
(prompt_tts_amd/encodec.py).

The reference builds `EncodecModel.encodec_model_24khz()` at import time, which downloads pretrained weights;
this build never fetches anything: pass `--weights <encodec state_dict .pt>` (original `encodec` package naming,
weight_g/weight_v are folded on load).  Without it a seeded random decoder is used (useful only for plumbing).
"""
from argparse import ArgumentParser

import numpy as np
import torch

from prompt_tts_amd.encodec import EncodecDecoder, weights_from_encodec_state_dict

_model = None


def load_decoder(weights_path=None, dtype=torch.bfloat16, device="cuda", seed=0):
    global _model
    if weights_path is not None:
        W = weights_from_encodec_state_dict(torch.load(weights_path, map_location="cpu"))
    else:
        W = random_decoder_weights(seed)
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
    model = _model if _model is not None else load_decoder()
    return model.decode(encoded_frames)


def main(args):
    codec_matrix = np.load(args.npy_path)
    encoded_frames = torch.tensor(codec_matrix)
    if len(codec_matrix.shape) != 3:
        encoded_frames = encoded_frames.unsqueeze(0)
    load_decoder(args.weights, torch.float32 if args.dtype == "f32" else torch.bfloat16)
    with torch.no_grad():
        wav_dec = decode(encoded_frames)
    from scipy.io import wavfile
    pcm = (wav_dec[0][0].clamp(-1, 1).cpu().numpy() * 32767.0).astype(np.int16)      # soundfile's default PCM_16
    wavfile.write(args.npy_path.replace(".npy", ".wav"), EncodecDecoder.sample_rate, pcm)


def parse_args():
    parser = ArgumentParser(description="Test converting codec codes back to waveform.")
    parser.add_argument("--npy_path", required=True, help="Path to codec codes matrix.")
    parser.add_argument("--weights", default=None, help="encodec state_dict (.pt); random decoder if omitted")
    parser.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    return parser.parse_args()


if __name__ == "__main__":
    main(parse_args())
