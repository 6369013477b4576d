#!/usr/bin/env python
"""Waveform -> codec codes, with the reference's surface (data_preparation/generate_code.py:18-103): `create_batch`,
`generate(batch)` and `python encode_codec.py --input_file x.tar` writing x_processed.tar with `<utt>.npy` (int64 [8, T]) and
`<utt>.len.txt` next to the text files.  The encoder + residual vector quantiser run on the MI355X kernels
(prompt_tts_amd/encodec.py: EncodecEncoder).

The reference builds `EncodecModel.encodec_model_24khz()` at import time (downloads pretrained weights) and reads audio with
torchaudio + encodec.utils.convert_audio; this build never fetches anything and has neither package: pass
`--weights <encodec state_dict .pt>` (original `encodec` package naming; weight_g / weight_v are folded on load; a seeded RANDOM
encoder only with an explicit `--random_weights`).  Audio members are PCM WAV files of any sample rate, 8/16/24/32-bit, read
with the stdlib `wave` module: stereo keeps its first channel (generate_code.py:26-27) and other rates are resampled to 24 kHz
with a windowed-sinc kernel (torchaudio.transforms.Resample defaults, which convert_audio applies: Hann window,
lowpass_filter_width 6, rolloff 0.99) -- host-side, once per file, like the reference.
"""
import io
import math
import tarfile
import wave
from argparse import ArgumentParser

import numpy as np
import torch

from prompt_tts_amd.encodec import EncodecEncoder, encoder_weights_from_encodec_state_dict

SAMPLE_RATE = 24000
_model = None


def random_encoder_weights(seed=1, n_q=8):
    """Seeded fan-in-scaled weights of the 24 kHz encoder architecture (no checkpoint is available offline)."""
    g = torch.Generator().manual_seed(seed)

    def t(*shape, fan=None):
        fan = fan or (shape[1] * (shape[2] if len(shape) > 2 else 1))
        return (torch.rand(shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5

    W = {"codebooks": torch.randn(n_q, 1024, 128, generator=g) * 0.5, "enc.conv0.w": t(32, 1, 7), "enc.conv0.b": t(32, fan=100)}
    C = 32
    for i, r in enumerate((2, 4, 5, 8)):
        W[f"enc.res{i}.c3.w"] = t(C // 2, C, 3); W[f"enc.res{i}.c3.b"] = t(C // 2, fan=100)
        W[f"enc.res{i}.c1.w"] = t(C, C // 2, 1); W[f"enc.res{i}.c1.b"] = t(C, fan=100)
        W[f"enc.res{i}.sc.w"] = t(C, C, 1); W[f"enc.res{i}.sc.b"] = t(C, fan=100)
        W[f"enc.down{i}.w"] = t(2 * C, C, 2 * r); W[f"enc.down{i}.b"] = t(2 * C, fan=100)
        C *= 2
    for l in range(2):
        W[f"enc.lstm.w_ih{l}"] = t(2048, 512); W[f"enc.lstm.w_hh{l}"] = t(2048, 512)
        W[f"enc.lstm.b_ih{l}"] = t(2048, fan=100); W[f"enc.lstm.b_hh{l}"] = t(2048, fan=100)
    W["enc.final.w"] = t(128, 512, 7); W["enc.final.b"] = t(128, fan=100)
    return W


def load_encoder(weights_path=None, dtype=torch.float32, device="cuda", seed=1, random_weights=False):
    global _model
    if weights_path is not None:
        W = encoder_weights_from_encodec_state_dict(torch.load(weights_path, map_location="cpu"))
    elif random_weights:
        W = random_encoder_weights(seed)
    else:
        raise RuntimeError("no Encodec weights: pass --weights / load_encoder(<encodec state_dict .pt>); codes from a random encoder "
                           "would silently become training data (random_weights=True for plumbing runs)")
    _model = EncodecEncoder(W, device=device, dtype=dtype)
    return _model


def resample(wav, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """(channels, n) float -> (channels, ceil(n * new / orig)): band-limited sinc interpolation with a Hann window, the
    algorithm and defaults of torchaudio.transforms.Resample (what encodec.utils.convert_audio applies, generate_code.py:28).
    torchaudio is not installed here: parity unpinned, checked by known answers (tests/test_host_cpu.py)."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return wav
    g = math.gcd(orig_freq, new_freq)
    o, n = orig_freq // g, new_freq // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = (torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx) * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)        # (n, 1, 2 width + o)
    length = wav.shape[-1]
    x = torch.nn.functional.pad(wav.reshape(-1, length).double(), (width, width + o))
    y = torch.nn.functional.conv1d(x[:, None], kernel, stride=o)                               # (channels, n, frames)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)[:, :int(math.ceil(n * length / o))]
    return y.to(wav.dtype).reshape(wav.shape[:-1] + (-1,))


def read_wav(fileobj, target_rate=SAMPLE_RATE):
    """PCM WAV -> float tensor (1, n) in [-1, 1) at 24 kHz: first channel of a multi-channel file (generate_code.py:26-27),
    resampled when the file's rate differs (generate_code.py:28)."""
    with wave.open(fileobj, "rb") as w:
        width, ch, rate = w.getsampwidth(), w.getnchannels(), w.getframerate()
        raw = w.readframes(w.getnframes())
    if width == 1:
        pcm = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        pcm = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        pcm = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        pcm = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError(f"unsupported PCM sample width {width} bytes")
    mono = torch.from_numpy(np.ascontiguousarray(pcm.reshape(-1, ch)[:, 0]))[None, :]
    return resample(mono, rate, target_rate)


def create_batch(members, tf, batch_size, max_duration):
    """Yield [wavs (1,1,L) padded to max_duration seconds, member names, code lengths ceil(n/320)] (generate_code.py:18-42)."""
    index = 1
    batch = [[], [], []]
    for member in members:
        if ".wav" not in member.name:
            continue
        wav = read_wav(io.BytesIO(tf.extractfile(member).read()))
        if wav.shape[1] > SAMPLE_RATE * max_duration:
            raise ValueError(f"{member.name} is longer than max_duration={max_duration}s")
        batch[1].append(member.name)
        batch[2].append(np.ceil(wav.shape[1] / 320))
        wav = torch.cat([wav, torch.zeros((1, SAMPLE_RATE * max_duration - wav.shape[1]))], dim=-1)
        batch[0].append(wav.unsqueeze(0))
        if index % batch_size == 0:
            yield batch
            batch = [[], [], []]
        index += 1
    if len(batch[0]) != 0:
        yield batch


def generate(batch):
    """list of (1, 1, L) waveforms -> numpy int64 codes (B, 8, L/320) (generate_code.py:45-51)."""
    if _model is None:
        raise RuntimeError("generate(): no encoder loaded -- call load_encoder(<encodec state_dict .pt>) first")
    wav = torch.cat(batch)
    return _model.encode(wav).cpu().numpy()


def _add_bytes(tar, name, payload):
    info = tarfile.TarInfo(name)
    info.size = len(payload)
    tar.addfile(info, io.BytesIO(payload))


def main(input_file, batch_size, max_duration):
    """x.tar -> x_processed.tar: per utterance `<utt>.npy` (int64 [8, T]) and `<utt>.len.txt`, then every text member
    unchanged at the end of the archive (the on-disk layout of generate_code.py:54-85; members are written from memory)."""
    out_path = input_file[:-4] + "_processed.tar" if input_file.endswith(".tar") else input_file + "_processed.tar"
    with tarfile.open(input_file, "r") as src, tarfile.open(out_path, "w") as dst:
        members = src.getmembers()
        for wavs, names, lengths in create_batch(members, src, batch_size, max_duration):
            for code, name, n_frames in zip(generate(wavs), names, lengths):
                stem = name.split("/")[-1].replace(".wav", "")
                buf = io.BytesIO()
                np.save(buf, code)
                _add_bytes(dst, stem + ".npy", buf.getvalue())
                _add_bytes(dst, stem + ".len.txt", str(n_frames).encode())
        for m in members:
            if m.isfile() and ".txt" in m.name:
                _add_bytes(dst, m.name.split("/")[-1], src.extractfile(m).read())
    return out_path


if __name__ == "__main__":
    cli = ArgumentParser(description="Waveforms -> Encodec codes (WebDataset-style tar in, tar out).")
    cli.add_argument("--input_file", type=str, required=True, help="tar with PCM .wav members of any sample rate (+ .txt transcripts)")
    cli.add_argument("--batch_size", type=int, default=32, help="waveforms per encode call")
    cli.add_argument("--max_duration", type=int, default=12, help="seconds every waveform is zero-padded to")
    cli.add_argument("--weights", type=str, default=None, help="encodec state_dict (.pt) in the original package's naming")
    cli.add_argument("--random_weights", action="store_true", help="seeded random encoder (meaningless codes: plumbing tests only)")
    cli.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ns = cli.parse_args()
    if ns.weights is None and not ns.random_weights:
        cli.error("--weights <encodec state_dict .pt> is required (or --random_weights for a plumbing run)")
    load_encoder(ns.weights, torch.float32 if ns.dtype == "f32" else torch.bfloat16, random_weights=ns.random_weights)
    print(main(ns.input_file, ns.batch_size, ns.max_duration))
