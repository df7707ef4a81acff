"""HE2RNA comparator on the HIP engine (sequoia_pub_amd/he2rna.py) against the golden vectors of the reference's own
class (tests/golden/he2rna.npz) and, at the size pretrain_gtex.py builds it, against the pinned oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sequoia_pub_amd  # noqa: E402,F401
from sequoia_pub_amd import _lib  # noqa: E402
from sequoia_pub_amd.he2rna import HE2RNA, fit  # noqa: E402
from oracle import he2rna_oracle as ho  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "he2rna.npz")


def _golden_model():
    d = np.load(GOLD)
    sd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w_")}
    D, G = sd["conv0.weight"].shape[1], sd["conv2.weight"].shape[0]
    m = HE2RNA(input_dim=D, output_dim=G, layers=[16, 16], ks=list(d["ks"]), dropout=0.5, device="cuda:0")
    m.load_state_dict(sd)
    return d, m, torch.from_numpy(d["x"]).cuda()


def test_forward_matches_reference_golden():
    _lib.require_gpu()
    d, m, x = _golden_model()
    m.eval()
    with torch.no_grad():
        ev = m(x).cpu().numpy()
    assert np.array_equal(np.isnan(ev), np.isnan(d["pred_eval"])) and np.isnan(ev[3]).all()      # 0/0 quirk reproduced
    np.testing.assert_allclose(ev[:3], d["pred_eval"][:3], rtol=1e-4, atol=1e-5)                 # north_star: 1e-4 relative, fp32
    for k in (1, 10, 100):
        with torch.no_grad():
            got = m.forward_fixed_k(x, k).cpu().numpy()
        ref = d["fixed_%d" % k]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        sc = m.conv(x).cpu()                                                                   # [B, G, N] like the reference's conv()
    ref_sc = ho.scores({k: v.cpu() for k, v in m.state_dict().items()}, x.cpu(), m.input_dim)
    np.testing.assert_allclose(sc.numpy(), ref_sc.numpy(), rtol=1e-4, atol=1e-5)


def test_gradients_match_reference_golden():
    _lib.require_gpu()
    d, m, x = _golden_model()
    m.eval()                                                   # dropout off, as in the golden run
    xg = x.clone().requires_grad_(True)
    (m.forward_fixed_k(xg, 20) * torch.from_numpy(d["r"]).cuda()).sum().backward()
    for name, p in m.named_parameters():
        ref = d["g_" + name]
        assert p.grad.shape == ref.shape
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), d["grad_x"], rtol=1e-4, atol=1e-4 * np.abs(d["grad_x"]).max())


def test_full_size_against_oracle_and_training_step():
    """pretrain_gtex.py:102-105: layers [256, 256], ks [1, 2, 5, 10, 20, 50, 100], 2048-dim cluster features, 20 820 genes."""
    _lib.require_gpu()
    torch.manual_seed(3)
    B, D, N, G = 6, 2048, 100, 20820
    m = HE2RNA(input_dim=D, output_dim=G, layers=[256, 256], ks=[1, 2, 5, 10, 20, 50, 100], device="cuda:0")
    x = torch.randn(B, N, D).clamp_min(-0.2)
    x[0, 80:] = 0
    xc = x.transpose(1, 2).contiguous()                        # channels x tiles, as the loops hand it over
    m.eval()
    with torch.no_grad():
        got = m(xc.cuda()).cpu()
        ref = ho.forward_eval({k: v.cpu() for k, v in m.state_dict().items()}, xc, m.ks, D)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=3e-3, weight_decay=0.0)
    y = torch.randn(B, G).cuda()
    before = m.conv2.weight.detach().clone()
    p1 = m(xc.cuda())
    loss = torch.nn.functional.mse_loss(p1, y)
    opt.zero_grad(); loss.backward(); opt.step()
    assert torch.isfinite(loss) and not torch.equal(before, m.conv2.weight.detach())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    with torch.no_grad():                                       # dropout is live in training mode
        a, b = m.forward_fixed_k(xc.cuda(), 10), m.forward_fixed_k(xc.cuda(), 10)
    assert not torch.equal(a, b)


def test_fit_loop_and_k_range(tmp_path):
    _lib.require_gpu()
    torch.manual_seed(0)
    np.random.seed(0)
    G, D, N = 12, 32, 20
    data = [(torch.rand(N, D), torch.rand(G), f"w{i}", "P") for i in range(12)]

    def coll(b):
        return torch.stack([t[0] for t in b]), torch.stack([t[1] for t in b]), np.array([t[2] for t in b]), np.array([t[3] for t in b])
    loader = torch.utils.data.DataLoader(data, batch_size=4, collate_fn=coll)
    m = HE2RNA(input_dim=D, output_dim=G, layers=[1], ks=[1, 5, 20], dropout=0.0, device="cuda:0")     # the reference's default hidden width of ONE
    preds, labels, wsis, projs = fit(m, 3e-3, loader, loader, loader, params={"max_epochs": 2, "patience": 5}, path=str(tmp_path), verbose=False)
    assert preds.shape == (12, G) and labels.shape == (12, G) and list(wsis[:2]) == ["w0", "w1"] and (preds >= 0).all()
    assert os.path.exists(os.path.join(str(tmp_path), "model.pt"))
    with pytest.raises(_lib.SequoiaHipError):                  # k > tiles: torch.topk raises in the reference
        HE2RNA(input_dim=D, output_dim=G, ks=[21], device="cuda:0").eval()(torch.rand(2, D, N).cuda())
