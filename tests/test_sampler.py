"""Reverse-diffusion sampler: scheduler arithmetic known-answer test on CPU; HIP sampler vs the oracle sampler
driven by the oracle denoiser on the GPU; the text -> codes -> waveform chain runs end to end."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_sampler_recovers_x0_when_eps_is_exact_cpu():
    """Known answer: with the exact noise as epsilon every step predicts x0, and the last step returns it."""
    from oracle.sampler import ddpm_sample
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(2, 4, 16, generator=g) * 1.6 - 0.8
    betas = torch.linspace(1e-4, 0.02, 1000).double(); ac = torch.cumprod(1 - betas, 0)
    for n_steps in (1000, 50, 7):
        def eps_fn(x, t):
            return ((x.double() - ac[t].sqrt() * x0.double()) / (1 - ac[t]).sqrt()).float()
        noises = [torch.randn(x0.shape, generator=g) for _ in range(n_steps)]
        out = ddpm_sample(eps_fn, torch.randn(x0.shape, generator=g), n_steps, noises)
        assert float((out - x0).abs().max()) < 1e-4


def test_step_coefficients_match_oracle_arithmetic_cpu():
    from prompt_tts_amd.sampler import schedule, step_coefficients, timesteps
    _, ac = schedule()
    ts, ratio = timesteps(50)
    assert ts[0] == 980 and ts[-1] == 0 and ratio == 20
    c_eps, c_inv, c_x0, c_xt, sigma = step_coefficients(0, -20, ac)
    assert abs(c_x0 - 1.0) < 1e-6 and abs(c_xt) < 1e-9 and sigma == 0.0
    c_eps, c_inv, c_x0, c_xt, sigma = step_coefficients(500, 480, ac)
    a_t, a_p = float(ac[500]), float(ac[480])
    assert abs(c_eps - (1 - a_t) ** 0.5) < 1e-6 and abs(c_inv - a_t ** -0.5) < 1e-5
    assert abs(sigma ** 2 - (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("n_steps", [6, 25])
def test_hip_sampler_vs_oracle(dev, n_steps):
    from oracle import model as om
    from oracle.init import deterministic_init_
    from oracle.sampler import ddpm_sample
    from prompt_tts_amd import sampler
    from prompt_tts_amd.tts.models import TTSSingleSpeaker
    z = np.load(os.path.join(GOLD, "model_small256.npz")); cfg = json.loads(str(z["config"]))
    seed = int(z["seed"])
    ref = deterministic_init_(om.TTSSingleSpeaker(cfg), seed).eval()
    m = deterministic_init_(TTSSingleSpeaker(cfg, dtype=torch.float32), seed).to(dev)
    ids = torch.from_numpy(z["ids"]); mask = torch.from_numpy(z["mask"])
    B, n_q, T = z["xt"].shape
    g = torch.Generator().manual_seed(5)
    x_init = torch.randn(B, n_q, T, generator=g); noises = [torch.randn(B, n_q, T, generator=g) for _ in range(n_steps)]
    with torch.no_grad():
        want = ddpm_sample(lambda x, t: ref(x, torch.full((B,), t, dtype=torch.int64), ids, mask).sample, x_init, n_steps, noises)
    got = sampler.sample(m, ids, mask, T, n_steps, x_init=x_init, noises=noises)
    assert got.shape == (B, n_q, T) and got.dtype == torch.float32
    err = float((got.cpu() - want).abs().max())
    assert err < 2e-3, err                                       # x lives in [-1, 1]: absolute == relative to full scale
    codes = sampler.synthesize(m, ids, mask, T, n_steps, x_init=x_init, noises=noises).cpu()
    from oracle.collate import denormalise_to_codes
    wc = torch.from_numpy(denormalise_to_codes(want.numpy()))
    assert codes.dtype == torch.int64 and int((codes - wc).abs().max()) <= 2          # a code step is 2/1023 ~ 2e-3
    assert float((codes == wc).float().mean()) > 0.7


@pytest.mark.gpu
def test_text_to_waveform_chain(dev):
    """ids -> DDPM sampling -> code rounding -> Encodec decode: shapes, ranges, determinism under injected noise."""
    from oracle import encodec as oe
    from oracle.init import deterministic_init_
    from prompt_tts_amd import sampler
    from prompt_tts_amd.encodec import EncodecDecoder
    from prompt_tts_amd.tts.models import TTSSingleSpeaker
    z = np.load(os.path.join(GOLD, "model_small256.npz")); cfg = json.loads(str(z["config"]))
    m = deterministic_init_(TTSSingleSpeaker(cfg, dtype=torch.bfloat16), int(z["seed"])).to(dev)
    dec = EncodecDecoder(oe.random_weights(3, n_q=cfg["in_channels"]), device=dev, dtype=torch.bfloat16)
    ids = torch.from_numpy(z["ids"]); mask = torch.from_numpy(z["mask"])
    B, n_q, T = z["xt"].shape
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    codes, wav = sampler.synthesize(m, ids, mask, T, 10, decoder=dec, generator=gen)
    assert codes.shape == (B, n_q, T) and int(codes.min()) >= 0 and int(codes.max()) <= 1023
    assert wav.shape == (B, 1, 320 * T) and bool(torch.isfinite(wav).all())
    gen.manual_seed(3)
    codes2, _ = sampler.synthesize(m, ids, mask, T, 10, decoder=dec, generator=gen)
    assert torch.equal(codes, codes2)
