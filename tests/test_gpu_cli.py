"""The four CLI counterparts end to end on a synthetic cohort (file formats of SURVEY section 8b):
patches store -> compute_features -> kmean_features -> main --train (k-fold) -> predict_independent_dataset."""
import os
import pickle

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kmeans_oracle, resnet_oracle  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib, store  # noqa: E402
from sequoia_pub_amd.cli import compute_features, kmean_features, main as train_main, predict_independent_dataset, pretrain_gtex  # noqa: E402
from sequoia_pub_amd.vis import ViS  # noqa: E402


def test_pipeline_clis(tmp_path):
    _lib.require_gpu()
    root = str(tmp_path)
    rs = np.random.RandomState(1)
    rows = []
    for i in range(10):
        slide = f"TCGA-AA-{i:04d}"
        d = os.path.join(root, "patches", slide)
        os.makedirs(d)
        f = store.File(os.path.join(d, slide + ".hdf5"), "w")
        for t in range(104 if i else 130):
            f.create_dataset(f"{t}_{t + 1}", data=rs.randint(0, 256, (224, 224, 3), dtype=np.uint8))
        f.close()
        rows.append(dict(wsi_file_name=slide, patient_id=f"P{i}", tcga_project="TCGA-BRCA",
                         **{f"rna_G{g}": float(rs.rand() * 6) for g in range(24)}))
    ref = os.path.join(root, "ref.csv")
    pd.DataFrame(rows).to_csv(ref, index=False)
    sd = resnet_oracle.init_resnet50_state_dict(seed=3)
    full = {**{k: v for k, v in __import__("sequoia_pub_amd.resnet", fromlist=["x"]).resnet50().state_dict().items()}, **sd}
    wpath = os.path.join(root, "resnet50.pth")
    torch.save(full, wpath)
    feat = os.path.join(root, "features")
    compute_features.main(["--feat_type", "resnet", "--ref_file", ref, "--patch_data_path", os.path.join(root, "patches"),
                           "--feature_path", feat, "--max_patch_number", "120", "--weights", wpath])
    f0 = store.File(os.path.join(feat, "TCGA-BRCA", "TCGA-AA-0000", "TCGA-AA-0000.h5"), "r")
    assert np.asarray(f0["resnet_features"][:]).shape == (120, 2048)            # random.sample to max_patch_number
    f0.close()
    f1 = store.File(os.path.join(feat, "TCGA-BRCA", "TCGA-AA-0001", "TCGA-AA-0001.h5"), "r")
    feats1 = np.asarray(f1["resnet_features"][:])
    assert feats1.shape == (104, 2048) and os.path.exists(os.path.join(feat, "TCGA-BRCA", "TCGA-AA-0001", "complete_tile.txt"))
    f1.close()                                     # HDF5 refuses "r+" while a read handle is open in the same process
    kmean_features.main(["--ref_file", ref, "--feature_path", feat, "--num_clusters", "100"])
    f1 = store.File(os.path.join(feat, "TCGA-BRCA", "TCGA-AA-0001", "TCGA-AA-0001.h5"), "r")
    cf = np.asarray(f1["cluster_features"][:])
    f1.close()
    o = kmeans_oracle.kmeans_fit(feats1)
    assert np.array_equal(cf, kmeans_oracle.cluster_means(feats1, o["labels"]))   # same labels, same fp32 means
    kmean_features.main(["--ref_file", ref, "--feature_path", feat])               # resume guard: nothing re-done
    train_main.main(["--ref_file", ref, "--feature_path", feat, "--save_dir", os.path.join(root, "exp"), "--exp_name", "t",
                     "--model_type", "vis", "--depth", "1", "--num-heads", "2", "--batch_size", "4", "--train", "--num_epochs", "2",
                     "--k", "5"])
    out = os.path.join(root, "exp", "TCGA", "t")
    res = pickle.load(open(os.path.join(out, "test_results.pkl"), "rb"))
    assert set(res["split_0"].keys()) == {"real", "preds", "random", "wsi_file_name", "tcga_project"} and len(res["genes"]) == 24
    assert res["split_0"]["preds"].shape == (2, 24)
    assert os.path.exists(os.path.join(out, "model_best.pt")) and os.path.exists(os.path.join(out, "model_best_4.pt"))
    assert os.path.exists(os.path.join(out, "train_0.npy"))
    # HF-format fold models + ensemble prediction
    for fold in range(2):
        m = ViS(24, 2048, 1, 2, 64, 64, 64, device="cpu")
        m.load_state_dict(torch.load(os.path.join(out, "model_best.pt" if fold == 0 else f"model_best_{fold}.pt")))
        m.save_pretrained(os.path.join(root, "hub", f"sequoia-brca-{fold}"))
    predict_independent_dataset.main(["--ref_file", ref, "--feature_path", feat, "--folds", "2", "--tcga_project", "TCGA-BRCA",
                                      "--depth", "1", "--num-heads", "2", "--save_dir", os.path.join(root, "pred"),
                                      "--model_dir", os.path.join(root, "hub")])
    tr = pickle.load(open(os.path.join(root, "pred", "exp", "test_results.pkl"), "rb"))
    assert tr["pred"].shape == (10, 24) and list(tr["pred"].columns) == [f"G{g}" for g in range(24)]
    # pre-training counterpart (src/pretrain_gtex.py): train phase only, best-loss checkpoint, same file name rule
    model, pdir = pretrain_gtex.main(["--path_csv", ref, "--feature_path", feat, "--save_dir", os.path.join(root, "pre"), "--exp_name", "g",
                                      "--quick", "1", "--batch_size", "4", "--compute_dtype", "bf16"])
    assert os.path.basename(pdir).endswith("_g") and os.path.exists(os.path.join(pdir, "model_best.pt"))
    sd = torch.load(os.path.join(pdir, "model_best.pt"), map_location="cpu")
    assert sd["linear_head.1.weight"].shape == (24, 2048) and "transformer.layers.5.1.net.3.weight" in sd
    # the MLP comparator through the same CLI (pretrain_gtex.py:102-105,118-120): whole model pickled as model.pt
    hmodel, hdir = pretrain_gtex.main(["--path_csv", ref, "--feature_path", feat, "--save_dir", os.path.join(root, "pre"), "--exp_name", "h",
                                       "--quick", "1", "--batch_size", "4", "--model", "he2rna"])
    assert os.path.exists(os.path.join(hdir, "model.pt")) and hmodel.conv2.weight.shape == (24, 256, 1)
