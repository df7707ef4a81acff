"""HE2RNA comparator (src/he2rna.py:42-106): the oracle against the golden vectors produced by the reference's own class."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import he2rna_oracle as ho  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "he2rna.npz")


def load():
    d = np.load(GOLD)
    sd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w_")}
    return d, sd, torch.from_numpy(d["x"])


def test_oracle_matches_reference_golden():
    d, sd, x = load()
    D = sd["conv0.weight"].shape[1]
    ev = ho.forward_eval(sd, x, d["ks"], D).numpy()
    assert np.array_equal(np.isnan(ev), np.isnan(d["pred_eval"])) and np.isnan(ev[3]).all()     # the 0/0 quirk is reproduced
    np.testing.assert_allclose(ev[:3], d["pred_eval"][:3], rtol=1e-6, atol=1e-6)
    for k in (1, 10, 100):
        got = ho.forward_fixed_k(sd, x, k, D).numpy()
        ref = d["fixed_%d" % k]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=1e-6, atol=1e-6)


def test_oracle_gradients_match_reference_golden():
    d, sd, x = load()
    D = sd["conv0.weight"].shape[1]
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xg = x.clone().requires_grad_(True)
    (ho.forward_fixed_k(p, xg, 20, D) * torch.from_numpy(d["r"])).sum().backward()
    for k, v in p.items():
        np.testing.assert_allclose(v.grad.numpy(), d["g_" + k], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xg.grad.numpy(), d["grad_x"], rtol=1e-5, atol=1e-6)
