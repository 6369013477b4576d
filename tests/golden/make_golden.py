"""Generate tests/golden/*.npz by running the UNMODIFIED reference modules in the authoring container.

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box -> never imported by tests; only the .npz travel).

The reference's own files (tts/models.py, tts/ldm/*.py) are imported as they lie.  They need a
handful of names from `diffusers`, which is not installed; those names are supplied from
oracle/blocks.py (the restated third-party arithmetic) plus inert mixins.  What this pins:
UNet topology, skip ordering, resnet/up/down-sample math, the positional-encoding quirk, the
skipped proj_out, the dead masks, and state_dict key names.  What it does NOT pin independently:
the diffusers block arithmetic (see tests/test_oracle_blocks.py for the torch.nn cross-checks).

Fixtures hold data only: config JSON, seeded inputs, outputs, per-key parameter checksums.
"""
import functools
import inspect
import json
import os
import sys
import types
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import blocks as ob  # noqa: E402
from oracle import model as om  # noqa: E402
from oracle.init import deterministic_init_  # noqa: E402


def _install_shim():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    d = mod("diffusers")
    d.__path__ = []
    cfg = mod("diffusers.configuration_utils")
    utils = mod("diffusers.utils")
    loaders = mod("diffusers.loaders")
    models = mod("diffusers.models")
    models.__path__ = []
    attn = mod("diffusers.models.attention")
    attnp = mod("diffusers.models.attention_processor")
    embs = mod("diffusers.models.embeddings")
    mutils = mod("diffusers.models.modeling_utils")

    def register_to_config(init):
        sig = inspect.signature(init)

        @functools.wraps(init)
        def wrapped(self, *args, **kwargs):
            bound = sig.bind(self, *args, **kwargs)
            bound.apply_defaults()
            vals = {k: v for k, v in bound.arguments.items() if k != "self"}
            object.__setattr__(self, "_cfg", SimpleNamespace(**vals))
            init(self, *args, **kwargs)
        return wrapped

    class ConfigMixin:
        @property
        def config(self):
            return self._cfg

    class ModelMixin(nn.Module):
        @property
        def dtype(self):
            return next(self.parameters()).dtype

    class BaseOutput(OrderedDict):
        def __init__(self, **kw):
            super().__init__(**kw)
            for k, v in kw.items():
                object.__setattr__(self, k, v)

    class _Logging:
        @staticmethod
        def get_logger(name):
            import logging
            return logging.getLogger(name)

    cfg.ConfigMixin, cfg.register_to_config = ConfigMixin, register_to_config
    utils.BaseOutput, utils.deprecate, utils.logging = BaseOutput, (lambda *a, **k: None), _Logging
    loaders.UNet2DConditionLoadersMixin = type("UNet2DConditionLoadersMixin", (), {})
    attn.BasicTransformerBlock = ob.BasicTransformerBlock
    attnp.AttentionProcessor, attnp.AttnProcessor = object, object
    embs.Timesteps, embs.TimestepEmbedding = ob.Timesteps, ob.TimestepEmbedding
    embs.GaussianFourierProjection = type("GaussianFourierProjection", (nn.Module,), {})
    mutils.ModelMixin = ModelMixin


def _synthetic(cfg, B, S, seed):
    g = torch.Generator().manual_seed(seed)
    n_q, T = cfg["in_channels"], cfg["sample_size"]
    code = torch.randint(0, 1024, (B, n_q, T), generator=g)
    x0 = ((code.double() / 1023.0).float() - 0.5) / 0.5
    noise = torch.randn(B, n_q, T, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    ids = torch.zeros(B, S, dtype=torch.int32)
    mask = torch.zeros(B, S, dtype=torch.int32)
    for b in range(B):
        L = int(torch.randint(S // 8, S // 2, (1,), generator=g))
        ph = torch.randint(1, 148, (L,), generator=g)
        seq = torch.full((2 * L + 1,), 148, dtype=torch.int64)
        seq[1::2] = ph
        n = min(S, seq.numel())
        ids[b, :n] = seq[:n].int()
        mask[b, :n] = 1
    return x0, noise, t, ids, mask


def _param_digest(sd):
    return {k: [list(v.shape), float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (authoring container only)")
    _install_shim()
    sys.path.insert(0, REF)
    from tts.models import TTSSingleSpeaker as RefModel  # the reference, unmodified

    cases = {
        # tiny config of the BASELINE family: both block kinds, skip concat, up/down-sample, cross-attn
        "tiny": (om.make_config(d=64, L=1, text_layers=1, n_q=2, T=64, S=32), 2, 32, 11),
        # two layers per block + different channel widths (exercises conv_shortcut on every up resnet)
        "wide": (dict(om.make_config(d=64, L=2, text_layers=2, n_q=8, T=128, S=48),
                      block_out_channels=[64, 128], attention_head_dim=32), 2, 48, 12),
        # GPU-runnable small cases (UNet head dim = C/8 must be 32/64/128 for the MI355X attention kernels)
        "small256": (om.make_config(d=256, L=1, text_layers=1, n_q=2, T=64, S=32), 2, 32, 21),
        "wide256": (dict(om.make_config(d=256, L=2, text_layers=2, n_q=8, T=128, S=48),
                         block_out_channels=[256, 512]), 2, 48, 22),
        # BASELINE config A at B=1 (d=256, 4 transformer layers, 2 codebooks, T=1024, S=256)
        "configA": (om.make_config(S=256, **om.CONFIG_A), 1, 256, 13),
    }
    for name, (cfg, B, S, seed) in cases.items():
        ref = deterministic_init_(RefModel(cfg).eval(), seed)
        sd = ref.state_dict()
        x0, noise, t, ids, mask = _synthetic(cfg, B, S, seed)
        xt = ob.add_noise(x0, noise, t)
        xt.requires_grad_(False)
        out = ref(xt, t, ids, mask).sample
        loss = torch.nn.functional.mse_loss(out.float(), noise.float())
        loss.backward()
        grads = {k: p.grad for k, p in ref.named_parameters()}
        unused = sorted(k for k, g in grads.items() if g is None)

        # the restatement must agree with the reference it restates before anything is written
        mine = om.TTSSingleSpeaker(cfg).eval()
        missing = mine.load_state_dict(sd, strict=True)
        out2 = mine(xt, t, ids, mask).sample
        err = float((out - out2).abs().max())
        assert err < 1e-5, (name, err)
        assert list(mine.state_dict().keys()) == list(sd.keys()), name

        pos = ref.text_encoder.pos_embedding(ref.text_encoder.word_embedding(ids.long()))[0]
        text_emb = ref.text_encoder(ids, mask)
        gsel = ["unet.conv_in.weight", "unet.conv_out.bias", "text_encoder.word_embedding.weight",
                "unet.mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight",
                "unet.up_blocks.1.resnets.0.conv_shortcut.weight", "unet.down_blocks.0.downsamplers.0.conv.weight"]
        np.savez_compressed(
            os.path.join(HERE, f"model_{name}.npz"),
            config=json.dumps(cfg), seed=seed, keys=json.dumps(list(sd.keys())),
            param_digest=json.dumps(_param_digest(sd)), unused=json.dumps(unused),
            x0=x0.numpy(), noise=noise.numpy(), t=t.numpy(), ids=ids.numpy(), mask=mask.numpy(), xt=xt.numpy(),
            out=out.detach().numpy(), loss=float(loss), pos=pos.detach().numpy(),
            text_emb=text_emb.detach().numpy(),
            **{"grad::" + k: grads[k].numpy() for k in gsel if k in grads and grads[k] is not None},
            grad_norm=float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None))),
        )
        nparam = sum(v.numel() for k, v in sd.items() if "inv_freq" not in k)
        print(f"{name}: params={nparam} out|max|={float(out.abs().max()):.4f} loss={float(loss):.6f} "
              f"oracle-vs-reference max abs err={err:.2e} unused={len(unused)}")


if __name__ == "__main__":
    main()
