"""Oracle (test infrastructure): host-side batch metrics of the train/evaluate loops.

Restates /root/reference/src/he2rna.py:140-149 (``compute_correlations``),
/root/reference/src/vit.py:32-33 (``smape``), and the two sklearn/torch metrics
the loops call: ``sklearn.metrics.mean_absolute_error`` (vit.py:167,272) and
``nn.MSELoss()`` (vit.py:129,166).
"""
import numpy as np


def compute_correlations(labels, preds):
    """he2rna.py:140-149: mean over genes of Pearson r(labels[:, g], preds[:, g]),
    skipping genes whose target column is constant and dropping NaN r."""
    metrics = []
    for i in range(labels.shape[1]):
        y_true = labels[:, i]
        if len(np.unique(y_true)) > 1:
            y_prob = preds[:, i]
            with np.errstate(invalid="ignore", divide="ignore"):
                metrics.append(np.corrcoef(y_true, y_prob)[0, 1])
    metrics = np.asarray(metrics)
    metrics = metrics[~np.isnan(metrics)]
    return np.mean(metrics)


def compute_correlations_vectorised(labels, preds):
    """Same quantity, column-vectorised in fp64 (used to time a fair CPU baseline
    and to check the device kernel on full-size [B, 20820] batches)."""
    y = labels.astype(np.float64)
    p = preds.astype(np.float64)
    yc = y - y.mean(0)
    pc = p - p.mean(0)
    syy = (yc * yc).sum(0)
    spp = (pc * pc).sum(0)
    syp = (yc * pc).sum(0)
    nonconst = (y != y[0:1]).any(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = syp / np.sqrt(syy * spp)
    r = r[nonconst]
    r = r[~np.isnan(r)]
    return r.mean()


def mean_absolute_error(y_true, y_pred):
    """sklearn.metrics.mean_absolute_error with default uniform multioutput
    averaging: mean over outputs of the per-output mean |err|."""
    return float(np.mean(np.mean(np.abs(y_pred.astype(np.float64) - y_true.astype(np.float64)), axis=0)))


def smape(A, F):
    """vit.py:32-33 (note: divides by len(A) = batch rows, sums over all entries)."""
    return 100 / len(A) * np.sum(2 * np.abs(F - A) / (np.abs(A) + np.abs(F)))


def mse(pred, target):
    """nn.MSELoss() default reduction='mean' over all B*G elements."""
    d = pred.astype(np.float64) - target.astype(np.float64)
    return float(np.mean(d * d))


# ---------------------------------------------------------------------------------------------
# test-set statistics: /root/reference/evaluation/evaluate_model.py:57-125 and
# /root/reference/evaluation/CorrelationStats.py:39-60 (dependent_corr, method='steiger')
# ---------------------------------------------------------------------------------------------
def dependent_corr_steiger(xy, xz, yz, n, twotailed=False):
    """CorrelationStats.py:50-60."""
    from scipy.stats import t
    d = xy - xz
    determin = 1 - xy * xy - xz * xz - yz * yz + 2 * xy * xz * yz
    av = (xy + xz) / 2
    cube = (1 - yz) * (1 - yz) * (1 - yz)
    t2 = d * np.sqrt((n - 1) * (1 + yz) / (((2 * (n - 1) / (n - 3)) * determin + av * av * cube)))
    p = 1 - t.cdf(abs(t2), n - 3)
    if twotailed:
        p *= 2
    return t2, p


def fdrcorrection(pvals):
    """statsmodels.stats.multitest.fdrcorrection(pvals) defaults (alpha 0.05, Benjamini-Hochberg, not sorted):
    returns the corrected p-values in the input order."""
    pvals = np.asarray(pvals, dtype=np.float64)
    order = np.argsort(pvals)
    ps = pvals[order]
    n = len(ps)
    raw = ps / (np.arange(1, n + 1) / float(n))
    corr = np.minimum.accumulate(raw[::-1])[::-1]
    corr[corr > 1] = 1
    out = np.empty_like(corr)
    out[order] = corr
    return out


def gene_eval_stats(real, pred, random):
    """evaluate_model.py:57-125 for one cancer type: per-gene loop exactly as written (scipy pearsonr, Steiger
    test one-tailed, RMSEs, quantile / mean normalised RMSE), then the NaN fills and the two FDR corrections.
    real/pred/random: [n, G].  Returns a dict of [G] arrays in GENE order (the reference then sorts by
    pred_real_r).  In the constant-column branch the reference assigns ``xy, xy, yz = 0, 0, 0`` (xz keeps the
    previous gene's value -- a typo); the intended xz = 0 is used here."""
    from scipy import stats
    n, G = real.shape
    keys = ("pred_real_r", "random_real_r", "pearson_p", "Steiger_p", "rmse_pred", "rmse_random", "rmse_quantile_norm", "rmse_mean_norm")
    out = {k: np.zeros(G) for k in keys}
    for g in range(G):
        r, p, z = real[:, g], pred[:, g], random[:, g]
        if len(set(p)) == 1 or len(set(r)) == 1 or len(set(z)) == 1:
            xy, xz, yz = 0, 0, 0
            p1, pst = 1, 1
        else:
            xy, p1 = stats.pearsonr(r, p)
            xz, _ = stats.pearsonr(r, z)
            yz, _ = stats.pearsonr(p, z)
            _, pst = dependent_corr_steiger(xy, xz, yz, len(r), twotailed=False)
        rmse_p = np.sqrt(np.mean((r.astype(np.float64) - p.astype(np.float64)) ** 2))      # mean_squared_error(squared=False)
        rmse_r = np.sqrt(np.mean((r.astype(np.float64) - z.astype(np.float64)) ** 2))
        out["pred_real_r"][g] = xy
        out["random_real_r"][g] = xz
        out["pearson_p"][g] = p1
        out["Steiger_p"][g] = pst
        out["rmse_pred"][g] = rmse_p
        out["rmse_random"][g] = rmse_r
        out["rmse_quantile_norm"][g] = rmse_p / (np.quantile(r, 0.75) - np.quantile(r, 0.25) + 1e-5)
        out["rmse_mean_norm"][g] = rmse_p / np.mean(r)
    out["pred_real_r"] = np.nan_to_num(out["pred_real_r"], nan=0.0)
    out["random_real_r"] = np.nan_to_num(out["random_real_r"], nan=0.0)
    out["pearson_p"] = np.nan_to_num(out["pearson_p"], nan=1.0)
    out["Steiger_p"] = np.nan_to_num(out["Steiger_p"], nan=1.0)
    out["fdr_pearson_p"] = fdrcorrection(out["pearson_p"])
    out["fdr_Steiger_p"] = fdrcorrection(out["Steiger_p"])
    return out
