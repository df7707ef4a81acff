"""CPU: the oracle's restatement of evaluation/evaluate_model.py:57-125 against the golden vectors produced with
the reference's own Steiger test (evaluation/CorrelationStats.py) and scipy on the same float32 tables."""
import os

import numpy as np

from oracle import metrics_oracle

KEYS = ("pred_real_r", "random_real_r", "pearson_p", "Steiger_p", "rmse_pred", "rmse_random", "rmse_quantile_norm", "rmse_mean_norm")


def test_oracle_matches_reference_statistics(golden_dir):
    z = np.load(os.path.join(golden_dir, "evalstats.npz"))
    out = metrics_oracle.gene_eval_stats(z["real"], z["pred"], z["random"])
    for k in KEYS:
        np.testing.assert_allclose(out[k], z[k], rtol=2e-5, atol=1e-7, err_msg=k)
    # constant columns take the reference's r = 0, p = 1 branch
    for g in (4, 11, 20):
        assert out["pred_real_r"][g] == 0 and out["pearson_p"][g] == 1 and out["Steiger_p"][g] == 1
    assert abs(out["pred_real_r"][30] - 1.0) < 1e-12 and out["rmse_pred"][30] == 0


def test_fdr_correction_is_benjamini_hochberg():
    p = np.array([0.01, 0.04, 0.03, 0.005, 0.5, 1.0])
    c = metrics_oracle.fdrcorrection(p)
    # hand-computed BH: sorted p * n / rank, then running minimum from the right
    np.testing.assert_allclose(c, [0.03, 0.06, 0.06, 0.03, 0.6, 1.0], rtol=1e-12)
    assert (c >= p).all() and (c <= 1).all()
