"""GPU: sq_gene_eval_stats + the host mirror of evaluate_model.py against the golden vectors and the oracle."""
import os

import numpy as np
import pytest

from oracle import metrics_oracle
from sequoia_pub_amd import _lib, evalstats, synth

pytestmark = pytest.mark.gpu
KEYS = ("pred_real_r", "random_real_r", "pearson_p", "Steiger_p", "rmse_pred", "rmse_random", "rmse_quantile_norm", "rmse_mean_norm")


def test_golden_tables(golden_dir):
    _lib.require_gpu()
    z = np.load(os.path.join(golden_dir, "evalstats.npz"))
    df = evalstats.gene_eval_stats(z["real"], z["pred"], z["random"])
    assert list(df["pred_real_r"]) == sorted(df["pred_real_r"], reverse=True)        # reference sorts by r, descending
    df = df.sort_index()
    for k in KEYS:
        ref = z[k]
        np.testing.assert_allclose(df[k].values, ref, rtol=2e-5, atol=1e-7, err_msg=k)
    ora = metrics_oracle.gene_eval_stats(z["real"], z["pred"], z["random"])
    for k in ("fdr_pearson_p", "fdr_Steiger_p"):
        np.testing.assert_allclose(df[k].values, ora[k], rtol=2e-5, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("n", [2, 3, 1000, 4097])
def test_full_gene_panel_vs_oracle_columns(n):
    """20 820 genes; odd and power-of-two-crossing sample counts exercise the padded bitonic sort."""
    _lib.require_gpu()
    G = 20820
    rs = np.random.RandomState(n)
    real = synth.rna_targets(7, n) if n <= 1000 else (rs.rand(n, G) * 8).astype(np.float32)
    real = real[:, :G]
    pred = (real * 0.5 + rs.randn(n, G).astype(np.float32)).astype(np.float32)
    rnd = (rs.rand(n, G) * 8).astype(np.float32)
    s, _ = evalstats.device_stats(real, pred, rnd)
    cols = rs.choice(G, 40, replace=False)
    r64, p64, z64 = real[:, cols].astype(np.float64), pred[:, cols].astype(np.float64), rnd[:, cols].astype(np.float64)
    np.testing.assert_allclose(s[3][cols], np.sqrt(((r64 - p64) ** 2).mean(0)), rtol=1e-10)
    np.testing.assert_allclose(s[4][cols], np.sqrt(((r64 - z64) ** 2).mean(0)), rtol=1e-10)
    np.testing.assert_allclose(s[5][cols], r64.mean(0), rtol=1e-12)
    np.testing.assert_allclose(s[6][cols], np.quantile(r64, 0.25, axis=0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(s[7][cols], np.quantile(r64, 0.75, axis=0), rtol=1e-12, atol=1e-12)
    if n > 2:
        with np.errstate(invalid="ignore", divide="ignore"):
            ref = np.array([np.corrcoef(r64[:, i], p64[:, i])[0, 1] for i in range(len(cols))])
        ok = ~np.isnan(ref)
        np.testing.assert_allclose(s[0][cols][ok], ref[ok], rtol=1e-9, atol=1e-12)


def test_rejects_oversized_or_mismatched_tables():
    _lib.require_gpu()
    a = np.zeros((9000, 8), np.float32)
    with pytest.raises(ValueError):
        evalstats.device_stats(a, a, a)
    with pytest.raises(ValueError):
        evalstats.device_stats(np.zeros((5, 8), np.float32), np.zeros((5, 9), np.float32), np.zeros((5, 8), np.float32))


def test_cli_writes_the_reference_csvs(tmp_path, golden_dir):
    """test_results.pkl layout of src/main.py:206-219 -> all_genes.csv / sig_genes.csv (evaluate_model.py:137-138)."""
    _lib.require_gpu()
    import pickle
    import pandas as pd
    from sequoia_pub_amd.cli import evaluate_model
    z = np.load(os.path.join(golden_dir, "evalstats.npz"))
    n = z["real"].shape[0]
    cut = [0, 30, 60, 90, 120, n]
    res = {"genes": [f"rna_G{i}" for i in range(z["real"].shape[1])]}
    for k in range(5):
        sl = slice(cut[k], cut[k + 1])
        res[f"split_{k}"] = {"real": z["real"][sl], "preds": z["pred"][sl], "random": z["random"][sl],
                             "wsi_file_name": [f"w{i}" for i in range(cut[k], cut[k + 1])]}
    os.makedirs(tmp_path / "brca")
    with open(tmp_path / "brca" / "test_results.pkl", "wb") as f:
        pickle.dump(res, f)
    evaluate_model.main(["--model_dir", str(tmp_path), "--cancers", "brca", "gbm"])
    allg = pd.read_csv(tmp_path / "results" / "all_genes.csv", index_col=0)
    sig = pd.read_csv(tmp_path / "results" / "sig_genes.csv", index_col=0)
    assert len(allg) == 60 and set(allg["cancer"]) == {"brca"}
    np.testing.assert_allclose(allg.loc["rna_G0", "pred_real_r"], z["pred_real_r"][0], rtol=2e-5)
    assert 0 < len(sig) < 60 and (sig["Steiger_p"] < 0.05).all() and "rna_G4" not in sig.index
