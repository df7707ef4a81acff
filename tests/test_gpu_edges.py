"""Edge cases the reference code handles (or trips over) on the path: empty batches in the loops, ragged slides in
the end-to-end pipeline, fine-tuning head replacement, save/load round trips."""
import os

import numpy as np
import pytest
import torch

from gpu_util import rel_err

pytestmark = pytest.mark.gpu

from oracle import kmeans_oracle, vis_oracle  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib, synth  # noqa: E402
from sequoia_pub_amd import train as sq_train  # noqa: E402
from sequoia_pub_amd.pipeline import SlidePipeline  # noqa: E402
from sequoia_pub_amd.resnet import resnet50  # noqa: E402
from sequoia_pub_amd.vis import ViS  # noqa: E402

CFG = dict(num_outputs=60, input_dim=128, depth=1, nheads=2, dimensions_f=64, dimensions_s=64, dimensions_c=64)


def _vis(seed=3, **over):
    cfg = dict(CFG, **over)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=seed), seed=seed + 1)
    m = ViS(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    return m.to("cuda:0"), sd


def test_loops_skip_empty_batches_like_the_reference():
    """custom_collate_fn returns [] when every sample of a batch lacks features (utils.py:10-18); train / evaluate /
    predict `continue` over those (vit.py:157-158, 262-263, 303-304)."""
    _lib.require_gpu()
    m, _ = _vis()
    g = torch.Generator().manual_seed(0)
    xs, ys = torch.randn(6, 100, 128, generator=g), torch.rand(6, 60, generator=g) * 8
    names = [f"w{i}" for i in range(6)]
    full = [(xs[i:i + 3], ys[i:i + 3], names[i:i + 3], ["P"] * 3) for i in (0, 3)]
    holes = [([], [], [], []), full[0], ([], [], [], []), full[1]]
    p_full, w_full, _ = sq_train.predict(m, full)
    p_holes, w_holes, _ = sq_train.predict(m, holes)
    assert np.array_equal(np.asarray(p_full), np.asarray(p_holes)) and list(w_full) == list(w_holes) == names
    e_full = sq_train.evaluate(m, full, verbose=False)
    e_holes = sq_train.evaluate(m, holes, verbose=False)
    assert np.array_equal(np.asarray(e_full[0]), np.asarray(e_holes[0]))


def test_pipeline_with_ragged_slides_equals_stage_by_stage():
    """Slides with different patch counts (one barely above n_clusters) through SlidePipeline == embed, oracle
    k-Means on the embedded features, ViS on the cluster means."""
    _lib.require_gpu()
    torch.manual_seed(11)
    rn = resnet50(pretrained=False, compute_dtype="fp32").to("cuda:0").eval()
    vis, sd = _vis(input_dim=2048)
    pipe = SlidePipeline(rn, vis, n_clusters=100, sub_batch=64)
    slides = [torch.from_numpy(synth.patches_u8(40 + i, n, 224)).cuda() for i, n in enumerate((101, 137))]
    out = pipe(slides)
    assert out["pred"].shape == (2, 60) and [len(l) for l in out["labels"]] == [101, 137]
    for i, s in enumerate(slides):
        f = rn.extract_patches_u8(s, sub_batch=500)
        assert torch.equal(f, out["features"][i])                 # sub-batching / streams do not change features
        r = kmeans_oracle.kmeans_fit(f.cpu().numpy())
        assert np.array_equal(r["labels"], out["labels"][i].cpu().numpy())
        cf = kmeans_oracle.cluster_means(f.cpu().numpy(), r["labels"])
        assert np.array_equal(cf, out["cluster_features"][i].cpu().numpy())
        with torch.no_grad():
            ref = vis_oracle.vis_forward(sd, torch.from_numpy(cf)[None]).numpy()[0]
        assert rel_err(out["pred"][i].cpu().numpy(), ref) < 1e-4


def test_head_replacement_and_checkpoint_round_trip(tmp_path):
    """main.py:155-157 replaces `linear_head` for fine-tuning; vit.py:213 saves `model.state_dict()` and
    predict_independent_dataset.py:75-80 loads it back."""
    _lib.require_gpu()
    m, sd = _vis()
    x = torch.randn(2, 100, 128, generator=torch.Generator().manual_seed(5)).cuda()
    m.linear_head = torch.nn.Sequential(torch.nn.LayerNorm(128), torch.nn.Linear(128, 17))
    m.to("cuda:0")
    out = m(x)
    assert out.shape == (2, 17)
    sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    assert sd2["linear_head.1.weight"].shape == (17, 128)
    with torch.no_grad():
        ref = vis_oracle.vis_forward(sd2, x.cpu()).numpy()
    assert rel_err(out.detach().cpu().numpy(), ref) < 1e-4
    path = os.path.join(tmp_path, "model_best.pt")
    torch.save(m.state_dict(), path)
    m2 = ViS(**dict(CFG, num_outputs=17), device="cuda:0")
    m2.load_state_dict(torch.load(path, map_location="cpu"))
    m2.to("cuda:0")
    assert torch.equal(m2(x), out)
    # the new head trains: one optimizer step changes it and leaves shapes alone
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    loss = torch.nn.functional.mse_loss(m(x), torch.zeros(2, 17, device="cuda"))
    loss.backward()
    opt.step()
    assert not torch.equal(m.state_dict()["linear_head.1.weight"].cpu(), sd2["linear_head.1.weight"])


def test_empty_inputs():
    """No patches / no slides: empty results, no launches, no exceptions; a slide with fewer patches than clusters is
    the caller's error, as with scikit-learn (kmean_features.py catches it per slide and moves on)."""
    _lib.require_gpu()
    rn = resnet50(pretrained=False, compute_dtype="bf16").to("cuda:0").eval()
    out = rn.extract_patches_u8(torch.empty(0, 224, 224, 3, dtype=torch.uint8).cuda())
    assert out.shape == (0, 2048)
    vis, _ = _vis(input_dim=2048)
    pipe = SlidePipeline(rn, vis)
    r = pipe([])
    assert r["pred"].shape == (0, 60) and r["labels"] == []
    with pytest.raises(Exception):
        pipe([torch.from_numpy(synth.patches_u8(1, 20, 224)).cuda()])      # 20 patches < 100 clusters


def test_streaming_submit_flush_equals_call():
    """submit()/flush() defer the last slide of every group to the next call; all results, in order, equal __call__'s."""
    _lib.require_gpu()
    torch.manual_seed(12)
    rn = resnet50(pretrained=False, compute_dtype="bf16").to("cuda:0").eval()
    vis, _ = _vis(input_dim=2048)
    pipe = SlidePipeline(rn, vis, sub_batch=64)
    slides = [torch.from_numpy(synth.patches_u8(60 + i, 110 + 7 * i, 224)).cuda() for i in range(5)]
    ref = pipe(slides)
    outs = []
    for group in (slides[:2], slides[2:3], slides[3:]):
        r = pipe.submit(group)
        if r is not None:
            outs.append(r)
    assert pipe.submit([]) is None or True
    r = pipe.flush()
    assert r is not None
    outs.append(r)
    assert pipe.flush() is None
    pred = torch.cat([o["pred"] for o in outs])
    labels = [l for o in outs for l in o["labels"]]
    assert pred.shape == ref["pred"].shape and torch.equal(pred, ref["pred"])
    assert all(torch.equal(a, b) for a, b in zip(labels, ref["labels"]))
