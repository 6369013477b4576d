"""Oracle restatement of ancestral DDPM sampling as diffusers 0.15 `DDPMScheduler.step` defines it (epsilon prediction,
linear betas, `fixed_small` variance clamped at 1e-20, `clip_sample=True`), driven by the oracle denoiser.  TEST INFRA ONLY.
parity unpinned: the reference contains no sampler; the scheduler arithmetic is recalled from the pinned
dependency that `train.py:32-36` instantiates."""
import torch


def ddpm_sample(eps_fn, x_init, n_steps, noises, n_train=1000, clip=1.0):
    """eps_fn(x, t_int) -> eps; x_init (B,n_q,T); noises: one tensor per step but the last.  f64 scheduler arithmetic."""
    betas = torch.linspace(1e-4, 0.02, n_train, dtype=torch.float32).double()
    ac = torch.cumprod(1.0 - betas, dim=0)
    ratio = n_train // n_steps
    x = x_init.double()
    for i in range(n_steps):
        t = (n_steps - 1 - i) * ratio
        prev = t - ratio
        eps = eps_fn(x.float(), t).double()
        a_t = ac[t]; a_prev = ac[prev] if prev >= 0 else torch.tensor(1.0, dtype=torch.float64)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_alpha = a_t / a_prev; cur_beta = 1 - cur_alpha
        x0 = ((x - b_t.sqrt() * eps) / a_t.sqrt()).clamp(-clip, clip)
        mean = a_prev.sqrt() * cur_beta / b_t * x0 + cur_alpha.sqrt() * b_prev / b_t * x
        if t > 0:
            var = (b_prev / b_t * cur_beta).clamp(min=1e-20)
            mean = mean + var.sqrt() * noises[i].double()
        x = mean
    return x.float()
