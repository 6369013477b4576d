"""Oracle restatement of one training step (train.py:79-120).  TEST INFRASTRUCTURE ONLY.

noise / timesteps are INPUTS (device RNG streams are never compared); everything else follows
the reference: x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps ; loss = mean((eps_hat-eps)^2) in fp32;
backward; clip global grad norm to 1.0; AdamW(lr 1e-5, betas (0.95,0.999), wd 1e-6, eps 1e-8);
LR factor from the schedule lambda.
"""
import torch
import torch.nn.functional as F

from .blocks import add_noise

ADAMW = dict(lr=1e-5, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)   # train.py:41-47
MAX_GRAD_NORM = 1.0                                                       # train.py:117


def make_optimizer(model):
    return torch.optim.AdamW(model.parameters(), **ADAMW)


def loss_and_grads(model, x0, noise, t, ids, mask):
    xt = add_noise(x0, noise, t)
    pred = model(xt, t, ids, mask).sample
    loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    model.zero_grad(set_to_none=True)
    loss.backward()
    return loss.detach(), pred.detach()


def train_step(model, optimizer, x0, noise, t, ids, mask, lr_factor=1.0):
    loss, _ = loss_and_grads(model, x0, noise, t, ids, mask)
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), MAX_GRAD_NORM)
    for g in optimizer.param_groups:
        g["lr"] = ADAMW["lr"] * lr_factor
    optimizer.step()
    optimizer.zero_grad()
    return float(loss), float(gnorm)
