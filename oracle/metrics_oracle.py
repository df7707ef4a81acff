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
