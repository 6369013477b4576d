"""CPU: the oracle against golden vectors produced by the unmodified reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model as om
from oracle.blocks import add_noise
from oracle.init import deterministic_init_

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["tiny", "wide", "small256", "wide256", "configA"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_fixture(name):
    z = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    cfg = json.loads(str(z["config"]))
    m = deterministic_init_(om.TTSSingleSpeaker(cfg).eval(), int(z["seed"]))
    sd = m.state_dict()
    assert list(sd.keys()) == json.loads(str(z["keys"]))                       # checkpoint contract: names AND order
    digest = json.loads(str(z["param_digest"]))
    for k, (shape, ssum, sabs) in digest.items():
        assert list(sd[k].shape) == shape
        assert abs(float(sd[k].double().sum()) - ssum) <= 1e-6 * max(1.0, sabs), k
    x0, noise, t = (torch.from_numpy(z[k]) for k in ("x0", "noise", "t"))
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    xt = add_noise(x0, noise, t)
    assert torch.equal(xt, torch.from_numpy(z["xt"]))
    out = m(xt, t, ids, mask).sample
    assert float((out - torch.from_numpy(z["out"])).abs().max()) < 1e-5
    loss = F.mse_loss(out.float(), noise.float())
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert sorted(k for k, g in grads.items() if g is None) == json.loads(str(z["unused"]))
    assert all("proj_out" in k for k in json.loads(str(z["unused"])))          # exactly the never-applied proj_out
    for key in z.files:
        if key.startswith("grad::"):
            g = grads[key[6:]]
            assert float((g - torch.from_numpy(z[key])).abs().max()) <= 1e-5 * max(1.0, float(g.abs().max())), key
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None))
    assert abs(float(gn) - float(z["grad_norm"])) < 1e-5 * float(z["grad_norm"])
    te = m.text_encoder
    pos = te.pos_embedding(te.word_embedding(ids.long()))[0]
    assert float((pos - torch.from_numpy(z["pos"])).abs().max()) < 1e-6
    assert float((te(ids, mask) - torch.from_numpy(z["text_emb"])).abs().max()) < 1e-5


def test_param_counts_match_survey():
    for c, want in ((om.CONFIG_A, 15.16e6), (om.CONFIG_B, 163.4e6)):
        m = om.TTSSingleSpeaker(om.make_config(**c))
        n = sum(p.numel() for p in m.parameters())
        assert abs(n - want) / want < 2e-3


def test_positional_quirk_closed_form():
    """pos[s,k] = sin(k w_{s//2}) (s even) / cos(k w_{s//2}) (s odd), w_j = 10000^(-2j/ch): features index position."""
    S, d, L = 10, 16, 12
    tab = om.positional_table(L, S, d)
    for s in (0, 1, 4, 7):
        w = 10000 ** (-2 * (s // 2) / 12)
        k = torch.arange(d, dtype=torch.float32)
        want = torch.sin(k * w) if s % 2 == 0 else torch.cos(k * w)
        assert torch.allclose(tab[s], want, atol=1e-6)
    with pytest.raises(RuntimeError):
        om.positional_table(8, 9, 4)


def test_bad_block_names_raise():
    cfg = om.make_config(d=64, L=1, text_layers=1, n_q=2, T=64, S=32)
    with pytest.raises(ValueError):
        om.TTSSingleSpeaker(dict(cfg, down_block_types=["Nope", "DownBlock1D"]))
    with pytest.raises(ValueError):
        om.TTSSingleSpeaker(dict(cfg, up_block_types=["UpBlock1D"]))
