"""Softmax ViT baseline (src/vit.py:49-115) on the HIP path vs reference golden vectors (tests/golden/vit_tiny.npz)
and the oracle at --model_type vit's real shape (dim 1024, 16 heads, mlp 2048)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import rel_err

pytestmark = pytest.mark.gpu

from oracle import vis_oracle  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib, synth  # noqa: E402
from sequoia_pub_amd import train as sq_train  # noqa: E402
from sequoia_pub_amd.vit import ViT  # noqa: E402


@pytest.mark.parametrize("mode,tol_f,tol_g", [("fp32", 1e-4, 1e-4), ("bf16", 3e-2, 8e-2)])
def test_vit_tiny_forward_and_grads(golden_dir, mode, tol_f, tol_g):
    _lib.require_gpu()
    z = np.load(os.path.join(golden_dir, "vit_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    m = ViT(num_outputs=40, dim=128, depth=2, heads=2, mlp_dim=256, device="cuda:0", compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda:0")
    x = torch.from_numpy(z["x"]).cuda().requires_grad_(True)
    y = torch.from_numpy(z["target"]).cuda()
    pred = m(x)
    e = rel_err(pred.detach().cpu().numpy(), z["pred"])
    print(f"vit_tiny {mode}: forward rel err {e:.3e}")
    assert e < tol_f
    loss, gpred = sq_train.mse_loss_grad(m, pred.detach(), y)
    pred.backward(gpred)
    gv = m.grad_views(m.flat.grad)
    worst = ("", 0.0)
    for k in (k for k in z.files if k.startswith("g::")):
        eg = rel_err(gv[k[3:]].cpu().numpy(), z[k])
        if eg > worst[1]:
            worst = (k, eg)
    print(f"vit_tiny {mode}: worst grad rel err {worst[1]:.3e} at {worst[0]}")
    assert worst[1] < tol_g, worst
    xo = torch.from_numpy(z["x"]).requires_grad_(True)
    torch.nn.functional.mse_loss(vis_oracle.vit_forward(sd, xo, 2), torch.from_numpy(z["target"])).backward()
    assert rel_err(x.grad.cpu().numpy(), xo.grad.numpy()) < tol_g


def test_vit_real_shape_vs_oracle():
    """main.py:160-163 shape (dim 1024, 16 heads x 64, mlp 2048), 2 layers, G = 500, B = 3."""
    _lib.require_gpu()
    torch.manual_seed(5)
    m = ViT(num_outputs=500, dim=1024, depth=2, heads=16, mlp_dim=2048, device="cuda:0")
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to("cuda:0")
    x = torch.from_numpy(synth.cluster_tokens(3, 3, 1024))
    with torch.no_grad():
        ref = vis_oracle.vit_forward(sd, x, 16).numpy()
        out = m(x.cuda()).cpu().numpy()
    assert rel_err(out, ref) < 1e-4
