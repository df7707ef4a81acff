"""The multi-rank INFERENCE paths on a 1-GPU box: N ranks share cuda:0, process group over gloo (the SQ_SHARE_GPU pattern of
test_gpu_ddp.py).  RCCL itself is the driver's multi-GPU run; here the point is that sharding changes WHO computes a row, never
the row: (a) BASELINE config 5's window sharding of ONE slide (spatial_vis/visualize.py:46-52: windows are independent) is
bit-identical to the one-rank result; (b) the reference's --start/--end slide sharding (compute_features_hdf5.py:80-85,
kmean_features.py:56-61) run as ranks through compute_features -> kmean_features -> predict_independent_dataset leaves the
same files as the one-rank runs, byte for byte, and test_results.pkl in the reference's row order
(predict_independent_dataset.py:65-96); (c) the visualize CLI over two ranks writes the one-rank CSV."""
import filecmp
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORKER = os.path.join(HERE, "multirank_worker.py")


def launch(nproc, args, timeout=900, extra_env=None):
    port = 29500 + (os.getpid() * 7 + abs(hash(tuple(args))) % 997) % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SQ_SHARE_GPU="1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), WORKER] + [str(a) for a in args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0
    return r


@pytest.mark.parametrize("case", [
    # nx, ny, mode, batch_windows, stride, holes, head_chunk, size, ranks
    (250, 200, "bf16", 1024, 1, 0, 4096, "full", 2),      # BASELINE config 5 AS STATED: 50 000 tiles, 47 769 windows, real model
    (23, 17, "fp32", 16, 1, 1, 64, "small", 2),           # holes, several head chunks per rank, ragged last batch and chunk
    (23, 17, "fp32", 16, 1, 1, 64, "small", 3),           # three ranks
    (31, 29, "fp32", 2, 10, 0, 4096, "small", 2),         # stride 10 (last writer wins); one head chunk: rank 1 owns no tile
    (23, 17, "bf16", 500, 3, 0, 100, "small", 2),         # one window batch: rank 1 runs no window
], ids=["config5_full_size_bf16", "holes_fp32", "three_ranks", "stride10_last_writer", "one_batch_stride3"])
def test_window_sharded_slide_is_bit_identical_to_the_one_rank_result(tmp_path, case):
    *args, ranks = case
    launch(ranks, ["spatial", tmp_path] + args)
    assert all((tmp_path / f"ok{r}").read_text() == "ok" for r in range(ranks))


def test_bench_spatial_two_ranks_is_window_sharded_with_strong_scaling():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SQ_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "spatial", "--grid", "60", "40", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "window-sharded x2"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]          # ONE slide per step for the whole job


def make_cohort(root, n=10, oversize=(0, 6)):
    from sequoia_pub_amd import store
    rs = np.random.RandomState(1)
    rows = []
    for i in range(n):
        slide = f"TCGA-AA-{i:04d}"
        d = os.path.join(root, "patches", slide)
        os.makedirs(d)
        f = store.File(os.path.join(d, slide + ".hdf5"), "w")
        for t in range(130 if i in oversize else 101 + i):
            f.create_dataset(f"{t}_{t + 1}", data=rs.randint(0, 256, (224, 224, 3), dtype=np.uint8))
        f.close()
        rows.append(dict(wsi_file_name=slide, patient_id=f"P{i}", tcga_project="TCGA-BRCA",
                         **{f"rna_G{g}": float(rs.rand() * 6) for g in range(24)}))
    ref = os.path.join(root, "ref.csv")
    pd.DataFrame(rows).to_csv(ref, index=False)
    return ref


def test_cli_chain_over_two_ranks_leaves_the_one_rank_files(tmp_path):
    from oracle import resnet_oracle       # checker only: the seeded weight recipe of the goldens
    from sequoia_pub_amd import _lib
    from sequoia_pub_amd.cli import compute_features, kmean_features, predict_independent_dataset
    from sequoia_pub_amd.data import shard_rows
    from sequoia_pub_amd.resnet import resnet50
    from sequoia_pub_amd.vis import ViS
    _lib.require_gpu()
    root = str(tmp_path)
    n = 10
    ref = make_cohort(root, n)
    wpath = os.path.join(root, "resnet50.pth")
    torch.save({**resnet50().state_dict(), **resnet_oracle.init_resnet50_state_dict(seed=3)}, wpath)
    torch.manual_seed(5)
    for fold in range(2):
        ViS(24, 2048, 1, 2, 64, 64, 64, device="cpu").save_pretrained(os.path.join(root, "hub", f"sequoia-brca-{fold}"))
    cf = ["--feat_type", "resnet", "--ref_file", ref, "--patch_data_path", os.path.join(root, "patches"), "--max_patch_number", "120",
          "--weights", wpath, "--compute_dtype", "f16x3"]
    km = ["--ref_file", ref, "--num_clusters", "100"]
    pr = ["--ref_file", ref, "--folds", "2", "--tcga_project", "TCGA-BRCA", "--depth", "1", "--num-heads", "2", "--model_dir", os.path.join(root, "hub")]
    # one rank, the reference's way of splitting the slide list: one process per --start/--end range (each seeds 99 afresh,
    # compute_features_hdf5.py:41,80-85), the ranges being the ones the two ranks take
    f1 = os.path.join(root, "features_1rank")
    for r in range(2):
        lo, hi = shard_rows(n, r, 2)
        compute_features.main(cf + ["--feature_path", f1, "--start", str(lo), "--end", str(hi)])
        kmean_features.main(km + ["--feature_path", f1, "--start", str(lo), "--end", str(hi)])
    predict_independent_dataset.main(pr + ["--feature_path", f1, "--save_dir", os.path.join(root, "pred1")])
    # two ranks under the launcher
    f2 = os.path.join(root, "features_2rank")
    launch(2, ["cli", "compute_features"] + cf + ["--feature_path", f2])
    launch(2, ["cli", "kmean_features"] + km + ["--feature_path", f2])
    launch(2, ["cli", "predict_independent_dataset"] + pr + ["--feature_path", f2, "--save_dir", os.path.join(root, "pred2")])
    for i in range(n):
        slide = f"TCGA-AA-{i:04d}"
        a, b = (os.path.join(f, "TCGA-BRCA", slide, slide + ".h5") for f in (f1, f2))
        assert filecmp.cmp(a, b, shallow=False), f"{slide}: the two-rank feature file differs from the one-rank file"
        assert filecmp.cmp(os.path.join(os.path.dirname(a), "complete_tile.txt"), os.path.join(os.path.dirname(b), "complete_tile.txt"), shallow=False)
    t1 = pickle.load(open(os.path.join(root, "pred1", "exp", "test_results.pkl"), "rb"))
    t2 = pickle.load(open(os.path.join(root, "pred2", "exp", "test_results.pkl"), "rb"))
    for key in ("pred", "random"):
        assert list(t2[key].index) == list(t1[key].index) == [f"TCGA-AA-{i:04d}" for i in range(n)]      # the reference's row order
        assert list(t2[key].columns) == list(t1[key].columns)
        a, b = t2[key].values, t1[key].values
        exact = np.array_equal(a, b)
        print(f"test_results.pkl['{key}']: two ranks (batches of 5 + 5 slides) vs one rank (one batch of 10): bit-equal {exact}, "
              f"max rel diff {float(np.abs(a - b).max() / np.abs(b).max()):.2e}")
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-5 * float(np.abs(b).max()))


def test_visualize_cli_over_two_ranks_writes_the_one_rank_csv(tmp_path):
    """spatial_vis/visualize.py:104-307 with the ONE slide dealt over two ranks (tile chunks for the feature cache, window
    batches and tile chunks for the aggregator): the CSV rank 0 writes equals the one-rank CSV byte for byte."""
    from oracle import resnet_oracle
    from sequoia_pub_amd import _lib
    from sequoia_pub_amd.cli import visualize
    from sequoia_pub_amd.resnet import resnet50
    from sequoia_pub_amd.vis import ViS
    _lib.require_gpu()
    root = str(tmp_path)
    rs = np.random.RandomState(4)
    nx, ny, G = 14, 13, 24
    arr = rs.randint(0, 256, ((ny + 1) * 256, (nx + 1) * 256, 3), dtype=np.uint8)
    os.makedirs(os.path.join(root, "TCGA", "P"))
    np.save(os.path.join(root, "TCGA", "P", "TCGA-X.npy"), arr)
    np.save(os.path.join(root, "mask.npy"), np.ones(((nx + 1) * 8, (ny + 1) * 8), dtype=bool))
    rw = os.path.join(root, "resnet.pth")
    torch.save({**resnet50().state_dict(), **resnet_oracle.init_resnet50_state_dict(seed=3)}, rw)
    ck = os.path.join(root, "vis_resnet", "st")
    os.makedirs(ck)
    pickle.dump({"genes": [f"G{i}" for i in range(G)]}, open(os.path.join(ck, "test_results.pkl"), "wb"))
    torch.manual_seed(7)
    for fold in (0, 1):
        torch.save(ViS(G, 2048, 6, 16, 64, 64, 64, device="cpu").state_dict(), os.path.join(ck, "model_best.pt" if fold == 0 else f"model_best_{fold}.pt"))
    common = ["--study", "st", "--project", "P", "--gene_names", "G3,G17", "--wsi_file_name", "TCGA-X.npy", "--save_folder", "t",
              "--feat_type", "resnet", "--slide_path", os.path.join(root, "TCGA", "P"), "--mask_path", os.path.join(root, "mask.npy"),
              "--extractor_weights", rw, "--compute_dtype", "fp32", "--model_type", "vis", "--folds", "0,1", "--checkpoint", ck]
    # --tile_chunk 64: the 182 valid tiles go through the extractor in three chunks (two for rank 0, one for rank 1)
    common += ["--tile_chunk", "64"]
    res, p1 = visualize.main(common + ["--out_root", os.path.join(root, "out1")])
    assert len(res) == nx * ny and np.isfinite(res["G3"].values).all()
    launch(2, ["cli", "visualize"] + common + ["--out_root", os.path.join(root, "out2")])
    p2 = p1.replace("out1", "out2")
    assert filecmp.cmp(p1, p2, shallow=False), "the two-rank CSV differs from the one-rank CSV"
