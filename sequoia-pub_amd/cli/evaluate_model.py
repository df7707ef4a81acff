"""Counterpart of /root/reference/evaluation/evaluate_model.py (whose paths are hard-coded): per cancer type read
``<model_dir>/<cancer>/test_results.pkl``, compute the per-gene statistics on the device, write
``<model_dir>/results/all_genes.csv`` and ``sig_genes.csv``.

    python -m sequoia_pub_amd.cli.evaluate_model --model_dir runs/ --cancers brca coad --folds 5
"""
import argparse
import os
import pickle

import pandas as pd

from ..evalstats import evaluate_test_results, significant_genes

CANCERS = ['brca', 'coad', 'gbm', 'kirp', 'kirc', 'luad', 'lusc', 'paad', 'prad', 'skcm', 'thca', 'ucec', 'hnsc', 'stad', 'blca', 'lihc']


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_dir", required=True)
    ap.add_argument("--folds", type=int, default=5)
    ap.add_argument("--cancers", nargs="*", default=CANCERS)
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    save_path = os.path.join(args.model_dir, "results")
    os.makedirs(save_path, exist_ok=True)
    df_list = []
    for cancer_type in args.cancers:
        path = os.path.join(args.model_dir, cancer_type, "test_results.pkl")
        if not os.path.exists(path):
            print(f"no data for {cancer_type}")             # evaluate_model.py:126-127
            continue
        print(cancer_type)
        with open(path, "rb") as f:
            test_res = pickle.load(f)
        df_list.append(evaluate_test_results(test_res, folds=args.folds, cancer_type=cancer_type, device=args.device))
    if not df_list:
        raise SystemExit("no test_results.pkl found under " + args.model_dir)
    all_res = pd.concat(df_list)
    all_res.to_csv(os.path.join(save_path, "all_genes.csv"))
    significant_genes(all_res).to_csv(os.path.join(save_path, "sig_genes.csv"))
    return all_res


if __name__ == "__main__":
    main()
