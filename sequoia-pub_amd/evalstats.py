"""Test-set statistics -- host-side mirror of /root/reference/evaluation/evaluate_model.py:28-139
(per cancer type: stack the folds of ``test_results.pkl``; per gene Pearson r of real vs predicted and real vs
random-model predictions with p-values, the Steiger test that the first exceeds the second
(``evaluation/CorrelationStats.py:39-60``), RMSEs and their normalised forms; Benjamini-Hochberg correction;
the significant-gene filter).

What moves to the device is the O(n x G) part -- the reference's loop of 20 820 x 3 ``scipy.stats.pearsonr`` calls,
RMSEs and quantiles -- as one ``sq_gene_eval_stats`` call; the p-values are closed-form functions of (r, n) and are
evaluated vectorised on the host with scipy's distributions."""
import numpy as np
import torch

from . import _lib

COLUMNS = ("pred_real_r", "random_real_r", "pearson_p", "Steiger_p", "rmse_pred", "rmse_random", "rmse_quantile_norm",
           "rmse_mean_norm")


def fdrcorrection(pvals):
    """statsmodels.stats.multitest.fdrcorrection defaults (Benjamini-Hochberg): corrected p-values, input order."""
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


def device_stats(real, pred, random, device="cuda:0"):
    """The sq_gene_eval_stats call: three [n, G] float tables -> double [9, G] (see include/sequoia_hip.h)."""
    _lib.require_gpu()
    t = [torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(device) if not torch.is_tensor(a)
         else a.to(device, torch.float32).contiguous() for a in (real, pred, random)]
    n, G = t[0].shape
    if t[1].shape != (n, G) or t[2].shape != (n, G):
        raise ValueError("real / pred / random must have the same [n, G] shape")
    need = _lib.lib().sq_gene_eval_workspace_bytes(n, G)
    if need == 0:
        raise ValueError(f"unsupported test-set size n={n} (2..8192 samples)")
    ws = torch.empty(need, dtype=torch.uint8, device=t[0].device)
    out = torch.empty(9, G, dtype=torch.float64, device=t[0].device)
    with torch.cuda.device(t[0].device):
        _lib.check(_lib.lib().sq_gene_eval_stats(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), n, G, _lib.ptr(out),
                                                 _lib.ptr(ws), need, _lib.stream_ptr(t[0].device)))
    return out.cpu().numpy(), n


def gene_eval_stats(real, pred, random, genes=None, device="cuda:0"):
    """evaluate_model.py:57-125 for one cancer type.  Returns a DataFrame indexed by gene with the reference's
    columns, sorted by ``pred_real_r`` descending, NaNs filled and the two FDR columns added."""
    import pandas as pd
    from scipy import stats
    s, n = device_stats(real, pred, random, device)
    const = s[8] > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        xy, xz, yz = s[0].copy(), s[1].copy(), s[2].copy()
        # scipy.stats.pearsonr two-sided p-value: r ~ Beta(n/2 - 1, n/2 - 1) on [-1, 1] under the null
        dist = stats.beta(n / 2 - 1, n / 2 - 1, loc=-1, scale=2)
        p1 = 2 * dist.cdf(-np.abs(xy))
        # CorrelationStats.dependent_corr(..., twotailed=False, method='steiger')
        d = xy - xz
        determin = 1 - xy * xy - xz * xz - yz * yz + 2 * xy * xz * yz
        av = (xy + xz) / 2
        cube = (1 - yz) ** 3
        t2 = d * np.sqrt((n - 1) * (1 + yz) / (((2 * (n - 1) / (n - 3)) * determin + av * av * cube)))
        pst = 1 - stats.t.cdf(np.abs(t2), n - 3)
        rmse_q, rmse_m = s[3] / (s[7] - s[6] + 1e-5), s[3] / s[5]
    xy[const], xz[const], p1[const], pst[const] = 0, 0, 1, 1          # the len(set(col)) == 1 branch
    df = pd.DataFrame({"pred_real_r": xy, "random_real_r": xz, "pearson_p": p1, "Steiger_p": pst,
                       "rmse_pred": s[3], "rmse_random": s[4], "rmse_quantile_norm": rmse_q, "rmse_mean_norm": rmse_m},
                      index=list(genes) if genes is not None else np.arange(s.shape[1]))
    df = df.sort_values("pred_real_r", ascending=False)
    df["pred_real_r"] = df["pred_real_r"].fillna(0)
    df["random_real_r"] = df["random_real_r"].fillna(0)
    df["pearson_p"] = df["pearson_p"].fillna(1)
    df["fdr_pearson_p"] = fdrcorrection(df["pearson_p"].values)
    df["Steiger_p"] = df["Steiger_p"].fillna(1)
    df["fdr_Steiger_p"] = fdrcorrection(df["Steiger_p"].values)
    return df


def evaluate_test_results(test_res, folds=5, cancer_type=None, device="cuda:0"):
    """evaluate_model.py:31-125: ``test_res`` is the dict ``src/main.py`` pickles (keys ``genes`` and
    ``split_{k}`` -> {'real', 'preds', 'random', 'wsi_file_name'})."""
    real = np.concatenate([np.asarray(test_res[f"split_{k}"]["real"]) for k in range(folds)])
    pred = np.concatenate([np.asarray(test_res[f"split_{k}"]["preds"]) for k in range(folds)])
    rnd = np.concatenate([np.asarray(test_res[f"split_{k}"]["random"]) for k in range(folds)])
    df = gene_eval_stats(real, pred, rnd, genes=test_res["genes"], device=device)
    if cancer_type is not None:
        df["cancer"] = cancer_type
    return df


def significant_genes(all_res):
    """evaluate_model.py:130-135."""
    return all_res[(all_res["pred_real_r"] > 0) & (all_res["pearson_p"] < 0.05) & (all_res["rmse_pred"] < all_res["rmse_random"]) &
                   (all_res["pred_real_r"] > all_res["random_real_r"]) & (all_res["Steiger_p"] < 0.05) &
                   (all_res["fdr_Steiger_p"] < 0.2)]
