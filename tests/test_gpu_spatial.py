"""sliding_window_method (spatial_vis/visualize.py:35-102) vs a literal CPU loop over the oracle model."""
import numpy as np
import pandas as pd
import pytest
import torch

from gpu_util import rel_err

pytestmark = pytest.mark.gpu

from oracle import vis_oracle  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib  # noqa: E402
from sequoia_pub_amd.spatial import enumerate_windows, sliding_window_all_genes, sliding_window_method  # noqa: E402
from sequoia_pub_amd.vis import ViS  # noqa: E402


def reference_loop(df, feats, sd, genes, stride, literal_2d):
    """visualize.py:35-102 restated with the oracle as the model (features come from the cache)."""
    max_x, max_y = max(df['xcoord_tf']), max(df['ycoord_tf'])
    preds = {g: {} for g in genes}
    for x in range(0, max_x, stride):
        for y in range(0, max_y, stride):
            window = df[((df['xcoord_tf'] >= x) & (df['xcoord_tf'] < (x + 10))) & ((df['ycoord_tf'] >= y) & (df['ycoord_tf'] < (y + 10)))]
            if window.shape[0] > 50:
                fa = feats[window.index.values]
                if fa.shape[0] < 100:
                    fa = torch.cat([fa, torch.zeros(100 - fa.shape[0], fa.shape[1])])
                with torch.no_grad():
                    out = vis_oracle.vis_forward(sd, fa[:, None, :] if literal_2d else fa[None])[0].numpy()
                for g in genes:
                    for key in window.index:
                        if stride == 10:
                            preds[g][key] = out[g]
                        else:
                            preds[g].setdefault(key, []).append(out[g])
    if stride < 10:
        for g in genes:
            for key in preds[g]:
                preds[g][key] = np.mean(preds[g][key])
    return preds


@pytest.mark.parametrize("stride,literal", [(10, False), (3, False), (1, False), (5, True)])
def test_sliding_window_matches_literal_loop(stride, literal):
    _lib.require_gpu()
    rs = np.random.RandomState(0)
    coords = [(x, y) for x in range(24) for y in range(19) if rs.rand() > 0.25]      # holes in the tissue mask
    df = pd.DataFrame(coords, columns=["xcoord_tf", "ycoord_tf"])
    feats = torch.from_numpy(rs.randn(len(df), 128).astype(np.float32))
    cfg = dict(num_outputs=40, input_dim=128, depth=1, nheads=2, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=5), seed=6)
    m = ViS(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    genes = [0, 7, 39]
    got = sliding_window_method(df, feats, m, genes, stride, literal_2d=literal, batch_windows=37)
    ref = reference_loop(df, feats, sd, genes, stride, literal)
    for g in genes:
        assert set(got[g].keys()) == set(ref[g].keys()) and len(ref[g]) > 100
        a = np.array([got[g][k] for k in sorted(ref[g])])
        b = np.array([ref[g][k] for k in sorted(ref[g])])
        assert rel_err(a, b) < 1e-4


def test_window_enumeration_rules():
    xs, ys = np.meshgrid(np.arange(12), np.arange(9), indexing="ij")
    members, origins = enumerate_windows(xs.ravel(), ys.ravel(), stride=1)
    # window at (0,0) holds 10*9 = 90 tiles (y only reaches 8), ascending df positions, -1 padded
    assert (members[0] >= 0).sum() == 90 and np.all(np.diff(members[0][:90]) > 0) and np.all(members[0][90:] == -1)
    assert all(((members[i] >= 0).sum() > 50) for i in range(len(members)))
    assert origins[:, 0].max() < 11 and origins[:, 1].max() < 8          # range(0, max, stride) excludes the max itself


@pytest.mark.parametrize("stride", [10, 2])
def test_all_gene_vote_matches_literal_loop(stride):
    """Config-5 form: every gene for every tile in one sq_window_vote; tiles no kept window covers are NaN."""
    _lib.require_gpu()
    rs = np.random.RandomState(3)
    coords = [(x, y) for x in range(21) for y in range(23) if rs.rand() > 0.3 or x > 14]
    df = pd.DataFrame(coords, columns=["xcoord_tf", "ycoord_tf"])
    feats = torch.from_numpy(rs.randn(len(df), 64).astype(np.float32))
    cfg = dict(num_outputs=37, input_dim=64, depth=1, nheads=1, dimensions_f=64, dimensions_s=64, dimensions_c=64)      # 37: scalar path
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=8), seed=9)
    m = ViS(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    out, votes = sliding_window_all_genes(df["xcoord_tf"].values, df["ycoord_tf"].values, feats, m, stride, batch_windows=50)
    out, votes = out.cpu().numpy(), votes.cpu().numpy()
    genes = list(range(37))
    ref = reference_loop(df, feats, sd, genes, stride, False)
    covered = sorted(ref[0].keys())
    assert set(np.nonzero(votes > 0)[0]) == set(covered) and 0 < len(covered)
    assert np.isnan(out[votes == 0]).all() and not np.isnan(out[votes > 0]).any()
    b = np.array([[ref[g][k] for g in genes] for k in covered])
    assert rel_err(out[covered], b) < 1e-4
    if stride < 10:
        assert votes.max() > 10                     # tiles in the interior belong to many windows


def reference_loop_probe_tiles(df, feats, sd, probes, batch=16):
    """visualize.py:35-102 at stride 1 restated for a SUBSET of tiles: every kept window that contains a probe tile goes
    through the oracle (all genes), a probe tile's prediction is the mean over its windows (visualize.py:96-100)."""
    max_x, max_y = max(df['xcoord_tf']), max(df['ycoord_tf'])
    px = {k: (int(df['xcoord_tf'][k]), int(df['ycoord_tf'][k])) for k in probes}
    wins = []
    for x in range(0, max_x):
        for y in range(0, max_y):
            if not any(x <= tx < x + 10 and y <= ty < y + 10 for tx, ty in px.values()):
                continue
            window = df[((df['xcoord_tf'] >= x) & (df['xcoord_tf'] < (x + 10))) & ((df['ycoord_tf'] >= y) & (df['ycoord_tf'] < (y + 10)))]
            if window.shape[0] > 50:
                wins.append(window.index.values)
    sums = {k: [] for k in probes}
    for i in range(0, len(wins), batch):
        chunk = wins[i:i + batch]
        x = torch.zeros(len(chunk), 100, feats.shape[1])
        for j, idx in enumerate(chunk):
            x[j, :len(idx)] = feats[idx]
        with torch.no_grad():
            out = vis_oracle.vis_forward(sd, x).numpy()
        for j, idx in enumerate(chunk):
            for k in probes:
                if k in idx:
                    sums[k].append(out[j])
    return {k: np.mean(np.stack(v), axis=0) for k, v in sums.items()}, len(wins)


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_config5_at_size_matches_literal_loop_on_probe_tiles(mode, tol):
    """BASELINE config 5 at the model's real size (D = 1024, depth 6, 16 heads, G = 20 820) on a 40 x 30 grid, through the
    path the bench runs: member gather inside the model's first kernel, vote before the head, one head product per tile,
    batch_windows = 1024 -- in fp32 and in bf16 (bf16 residual stream in inference).  Checked on probe tiles (corners,
    edges, interior: 1 ... 100 windows each) against the literal per-window loop over the oracle model."""
    _lib.require_gpu()
    nx, ny = 40, 30
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    df = pd.DataFrame({"xcoord_tf": xs.ravel(), "ycoord_tf": ys.ravel()})
    feats = torch.randn(nx * ny, 1024, generator=torch.Generator().manual_seed(11))
    cfg = dict(num_outputs=20820, input_dim=1024, depth=6, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=12), seed=13)
    m = ViS(**cfg, device="cuda:0", compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    out, votes = sliding_window_all_genes(df["xcoord_tf"].values, df["ycoord_tf"].values, feats.cuda(), m, 1, batch_windows=1024)
    out, votes = out.cpu().numpy(), votes.cpu().numpy()
    at = lambda x, y: x * ny + y
    probes = [at(0, 0), at(39, 29), at(20, 15), at(3, 27), at(39, 4), at(12, 0)]
    ref, n_windows = reference_loop_probe_tiles(df, feats, sd, probes)
    assert out.shape == (nx * ny, 20820) and not np.isnan(out).any()
    errs = {k: rel_err(out[k], ref[k]) for k in probes}
    print(f"config 5 at size, {mode}: {n_windows} oracle windows, votes of the probe tiles {[int(votes[k]) for k in probes]}, "
          f"worst rel err {max(errs.values()):.2e}")
    assert votes[at(0, 0)] == 1 and votes[at(20, 15)] == 100 and votes.max() == 100
    assert max(errs.values()) < tol, errs


def test_config5_full_size_50k_tiles_probe_tiles_and_batch_invariance():
    """BASELINE config 5 AS STATED: 50 000 tiles on a 250 x 200 grid, all valid -> 47 769 windows of 100 tokens at stride 1, the model
    at its real size, bf16, 2048 windows per forward (what bench.py --workload spatial times).  (a) Probe tiles -- a corner (1
    window), an edge and an interior tile (100 windows) -- against the literal per-window loop of visualize.py:35-102 over the
    oracle model; (b) the whole [50 000, 20 820] result must not depend on how the windows are batched (2048 vs a ragged 700):
    every window's rows go through the same arithmetic whatever its neighbours in the batch are."""
    _lib.require_gpu()
    nx, ny = 250, 200
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    df = pd.DataFrame({"xcoord_tf": xs.ravel(), "ycoord_tf": ys.ravel()})
    feats = torch.randn(nx * ny, 1024, generator=torch.Generator().manual_seed(21))
    cfg = dict(num_outputs=20820, input_dim=1024, depth=6, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=22), seed=23)
    m = ViS(**cfg, device="cuda:0", compute_dtype="bf16")
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    fd = feats.cuda()
    out, votes = sliding_window_all_genes(df["xcoord_tf"].values, df["ycoord_tf"].values, fd, m, 1, batch_windows=2048)
    assert out.shape == (nx * ny, 20820) and bool(torch.isfinite(out[votes > 0]).all())
    assert int(votes.max()) == 100 and int((votes > 0).sum()) > 49000
    at = lambda x, y: x * ny + y
    probes = [at(0, 0), at(120, 0), at(131, 97)]
    ref, n_windows = reference_loop_probe_tiles(df, feats, sd, probes)
    errs = {k: rel_err(out[k].cpu().numpy(), ref[k]) for k in probes}
    print(f"config 5 at FULL size (50 000 tiles), bf16: {n_windows} oracle windows for {len(probes)} probe tiles, votes {[int(votes[k]) for k in probes]}, "
          f"worst rel err {max(errs.values()):.2e}")
    assert int(votes[at(0, 0)]) == 1 and int(votes[at(131, 97)]) == 100
    assert max(errs.values()) < 3e-2, errs
    out2, votes2 = sliding_window_all_genes(df["xcoord_tf"].values, df["ycoord_tf"].values, fd, m, 1, batch_windows=700)
    assert torch.equal(votes, votes2)
    keep = votes > 0
    diff = float((out[keep] - out2[keep]).abs().max() / out[keep].abs().max())
    print(f"  2048 vs 700 windows per forward: max difference {diff:.2e} of max")
    assert diff < 1e-5


def test_first_layer_projection_per_tile_matches_the_per_token_product():
    """bf16 sliding-window path: layer 0's local projection taken once per TILE and gathered per window token
    (ViS.tile_projections + sq_vis_forward_tiles: f(tile feature + position) = f_tile[tile] + f_pos[slot]) against the per-token
    product of sq_vis_forward_ex on the same windows -- holes (zero-padded windows), the model at its real size.  The two differ by
    where the bf16 rounding of the operand happens (feature and position separately vs their sum): bf16-level, far inside the
    mode's distance from the fp32 oracle (5e-3)."""
    _lib.require_gpu()
    from sequoia_pub_amd.spatial import sliding_window_all_genes_sharded
    nx, ny = 36, 25
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    keep = np.random.default_rng(3).random(nx * ny) < 0.9
    x, y = xs.ravel()[keep], ys.ravel()[keep]
    cfg = dict(num_outputs=20820, input_dim=1024, depth=6, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=41), seed=42)
    m = ViS(**cfg, device="cuda:0", compute_dtype="bf16")
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    feats = torch.randn(x.size, 1024, generator=torch.Generator().manual_seed(43)).cuda()
    a, _, va = sliding_window_all_genes_sharded(x, y, feats, m, 1, batch_windows=256, tile_projection=True)
    b, _, vb = sliding_window_all_genes_sharded(x, y, feats, m, 1, batch_windows=256, tile_projection=False)
    assert torch.equal(va, vb) and int((va > 0).sum()) > 0.9 * x.size
    cov = va > 0
    diff = float((a[cov] - b[cov]).abs().max() / b[cov].abs().max())
    # the projections themselves against torch on the joined operands
    f_tile, f_pos = m.tile_projections(feats)
    wf = torch.cat([sd[f"transformer.layers.0.0.mixers.{h}.f.weight"] for h in range(16)]).cuda()
    bf = torch.cat([sd[f"transformer.layers.0.0.mixers.{h}.f.bias"] for h in range(16)]).cuda()
    ref_tile = feats.bfloat16().double() @ wf.bfloat16().double().T
    ref_pos = sd["pos_emb1D"].cuda().bfloat16().double() @ wf.bfloat16().double().T + bf.double()
    e_t = float((f_tile.double() - ref_tile).abs().max() / ref_tile.abs().max())
    e_p = float((f_pos.double() - ref_pos).abs().max() / ref_pos.abs().max())
    print(f"layer-0 projection per tile vs per window token ({int(cov.sum())} covered tiles, bf16): max difference {diff:.2e} of max; "
          f"f_tile vs fp64 on the bf16 operands {e_t:.1e}, f_pos {e_p:.1e}")
    assert e_t < 1e-5 and e_p < 1e-5
    assert diff < 5e-3


def test_visualize_cli_on_a_synthetic_slide(tmp_path):
    """spatial_vis/visualize.py:104-307 end to end: an in-memory 20x slide, mask -> valid tiles -> ResNet feature cache ->
    two-fold ViS ensemble and one HE2RNA fold -> stride-1 CSV; the CSV equals the library calls on the same cache."""
    import os
    import pickle
    from oracle import resnet_oracle
    from sequoia_pub_amd.cli import visualize
    from sequoia_pub_amd.he2rna import HE2RNA
    from sequoia_pub_amd.resnet import resnet50
    from sequoia_pub_amd.vis import ViS
    _lib.require_gpu()
    root = str(tmp_path)
    rs = np.random.RandomState(4)
    nx, ny, G = 9, 8, 24
    arr = rs.randint(0, 256, ((ny + 1) * 256, (nx + 1) * 256, 3), dtype=np.uint8)         # [height, width, 3]
    os.makedirs(os.path.join(root, "TCGA", "P"))
    np.save(os.path.join(root, "TCGA", "P", "TCGA-X.npy"), arr)
    mask = np.ones(((nx + 1) * 8, (ny + 1) * 8), dtype=bool)                                # slide width x height / 32, all tissue
    mask[:, 56:] = False                                                                   # background from tile row 7 on
    np.save(os.path.join(root, "mask.npy"), mask)
    genes = [f"G{i}" for i in range(G)]
    rw = os.path.join(root, "resnet.pth")
    rn = resnet50()
    torch.save({**rn.state_dict(), **resnet_oracle.init_resnet50_state_dict(seed=3)}, rw)
    for mt in ("vis", "he2rna"):
        ck = os.path.join(root, f"{mt}_resnet", "st")
        os.makedirs(ck)
        pickle.dump({"genes": genes}, open(os.path.join(ck, "test_results.pkl"), "wb"))
        torch.manual_seed(7)
        for fold in (0, 1):
            if mt == "vis":
                m = ViS(G, 2048, 6, 16, 64, 64, 64, device="cpu")
                torch.save(m.state_dict(), os.path.join(ck, "model_best.pt" if fold == 0 else f"model_best_{fold}.pt"))
            else:
                m = HE2RNA(input_dim=2048, layers=[256, 256], ks=[1, 2, 5, 10, 20, 50, 100], output_dim=G)
                torch.save(m, os.path.join(ck, f"model_{fold}.pt"))
    common = ["--study", "st", "--project", "P", "--gene_names", "G3,G17,nope", "--wsi_file_name", "TCGA-X.npy", "--save_folder", "t",
              "--feat_type", "resnet", "--slide_path", os.path.join(root, "TCGA", "P"), "--mask_path", os.path.join(root, "mask.npy"),
              "--out_root", os.path.join(root, "vis_out"), "--extractor_weights", rw, "--compute_dtype", "fp32"]
    res, path = visualize.main(common + ["--model_type", "vis", "--folds", "0,1", "--checkpoint", os.path.join(root, "vis_resnet", "st")])
    assert os.path.basename(path) == "stride-1.csv" and len(res) == nx * (ny - 1)            # the background strip drops one tile row
    assert {"xcoord", "ycoord", "xcoord_tf", "ycoord_tf", "G3_0", "G3_1", "G3", "G17"} <= set(res.columns) and "nope" not in res.columns
    back = pd.read_csv(path, index_col=0)
    np.testing.assert_allclose(back["G3"].values, res[["G3_0", "G3_1"]].mean(axis=1).values, rtol=1e-6)
    # the same numbers from the library on the same feature cache
    feat_model = resnet50()
    feat_model.load_state_dict(torch.load(rw))
    feat_model = feat_model.to("cuda:0").eval()
    df = visualize.valid_tiles(mask, (arr.shape[1], arr.shape[0]), 256)
    tiles = visualize.read_tiles(visualize.open_slide(os.path.join(root, "TCGA", "P", "TCGA-X.npy")), df, 256)
    cache = feat_model.extract_patches_u8(tiles.cuda())
    m = ViS(G, 2048, 6, 16, 64, 64, 64, device="cuda:0")
    m.load_state_dict(torch.load(os.path.join(root, "vis_resnet", "st", "model_best_1.pt")))
    direct = sliding_window_method(df, cache, m.to("cuda:0").eval(), [3], 1)
    np.testing.assert_allclose(res["G3_1"].values, np.array([direct[3][i] for i in res.index]), rtol=1e-5, atol=1e-6)
    res_h, _ = visualize.main(common + ["--model_type", "he2rna", "--folds", "1", "--checkpoint", os.path.join(root, "he2rna_resnet", "st")])
    assert np.isfinite(res_h["G17"].values).all() and np.array_equal(res_h["G17"].values, res_h["G17_1"].values)
