"""Reverse diffusion for TTSSingleSpeaker: the inference procedure the reference implies (it trains an epsilon-predictor
against `DDPMScheduler(num_train_timesteps=1000)`, train.py:32-36,96-107) but never wrote.

Ancestral DDPM sampling with the scheduler's defaults (linear betas 1e-4..0.02, epsilon prediction, `fixed_small` variance,
`clip_sample=True` at 1.0), optionally on a strided subset of the 1000 steps (`set_timesteps(n)`: t = (n-1-i) * (1000 // n)),
then the inverse of the collate normalisation (a-9: code = round((x + 1) / 2 * 1023)) and, if a decoder is given, Encodec
decode.  The per-step update is one elementwise kernel (pt_ddpm_step); the denoiser call is the model's forward.
"""
import torch

from . import engine as E
from . import ops


def schedule(n_train=1000, beta_start=1e-4, beta_end=0.02):
    betas = torch.linspace(beta_start, beta_end, n_train, dtype=torch.float32)
    return betas, torch.cumprod(1.0 - betas, dim=0)


def step_coefficients(t, prev_t, alphas_cumprod):
    """(c_eps, c_inv, c_x0, c_xt, sigma) of DDPMScheduler.step for the move t -> prev_t (prev_t < 0: the final step)."""
    ac = alphas_cumprod.double()
    a_t = float(ac[t]); a_prev = float(ac[prev_t]) if prev_t >= 0 else 1.0
    b_t, b_prev = 1.0 - a_t, 1.0 - a_prev
    cur_alpha = a_t / a_prev; cur_beta = 1.0 - cur_alpha
    c_x0 = a_prev ** 0.5 * cur_beta / b_t
    c_xt = cur_alpha ** 0.5 * b_prev / b_t
    var = max(b_prev / b_t * cur_beta, 1e-20)
    return b_t ** 0.5, 1.0 / a_t ** 0.5, c_x0, c_xt, (var ** 0.5 if t > 0 else 0.0)


def timesteps(n_steps, n_train=1000):
    if not 1 <= n_steps <= n_train:
        raise ValueError("1 <= n_steps <= 1000")
    ratio = n_train // n_steps
    return [(n_steps - 1 - i) * ratio for i in range(n_steps)], ratio


@torch.no_grad()
def sample(model, text_seq_ids, attention_mask, T, n_steps=1000, generator=None, x_init=None, noises=None, clip=1.0):
    """-> x0 (B, n_q, T) f32 in [-1, 1].  `x_init` / `noises` (list, one (B,n_q,T) tensor per step but the last) inject the
    random draws (tests never compare device RNG streams); otherwise `generator` (device) draws them."""
    st = model.store
    dev = st.device
    B, n_q = text_seq_ids.shape[0], model.config["in_channels"]
    x = x_init.to(dev, torch.float32).clone() if x_init is not None else torch.randn(B, n_q, T, device=dev, generator=generator)
    _, ac = schedule()
    ts, ratio = timesteps(n_steps)
    ids = text_seq_ids.to(device=dev, dtype=torch.int32).contiguous()
    mask = attention_mask.to(dev) if attention_mask is not None else None
    S, cpad = ids.shape[1], model.unet.cpad
    # the conditioning does not change over the reverse steps: the text encoder runs ONCE and every cross-attention layer
    # projects its K / V once (E.cross_kv_cache), instead of 1000 times each
    st.ensure_shadow_fresh()
    text_emb, _ = model.text_encoder.fwd(st, ids, mask, B, S)
    with E.cross_kv_cache():
        for i, t in enumerate(ts):
            xt = torch.empty(B * T, cpad, dtype=st.dtype, device=dev)
            ops.bct_to_tokens(x, xt, B, n_q, T, cpad)
            pred, _ = model.unet.fwd(st, xt, torch.full((B,), t, dtype=torch.int64, device=dev), text_emb, B, T, S)
            eps = torch.empty(B, n_q, T, dtype=torch.float32, device=dev)
            ops.tokens_to_bct(pred, eps, B, n_q, T, cpad)
            c_eps, c_inv, c_x0, c_xt, sigma = step_coefficients(t, t - ratio, ac)
            z = None
            if t > 0:
                z = noises[i].to(dev, torch.float32) if noises is not None else torch.randn(x.shape, device=dev, generator=generator)
            out = torch.empty_like(x)
            ops.ddpm_step(x, eps, z, out, c_eps, c_inv, clip, c_x0, c_xt, sigma)
            x = out
    return x


@torch.no_grad()
def synthesize(model, text_seq_ids, attention_mask, T, n_steps=1000, decoder=None, **kw):
    """text ids -> codes (B, n_q, T) int64 [-> waveform (B, 1, 320 T) when an EncodecDecoder is given]."""
    x0 = sample(model, text_seq_ids, attention_mask, T, n_steps, **kw)
    codes = ops.codes_from_continuous(x0)
    return (codes, decoder.decode(codes)) if decoder is not None else codes
