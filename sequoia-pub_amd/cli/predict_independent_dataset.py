"""5-fold ensemble inference -- counterpart of /root/reference/evaluation/predict_independent_dataset.py
(same flags; writes ``<save_dir>/<exp_name>/test_results.pkl`` = {'pred': DataFrame, 'random': DataFrame}).
``--model_dir`` (extra) points at local ``sequoia-{cancer}-{fold}`` HuggingFace folders (config.json +
model.safetensors) when the hub is unreachable.  Under torchrun the slide list is sharded across GPUs and
gathered on rank 0."""
import argparse
import os
import pickle

import numpy as np
import pandas as pd
import torch
from torch.utils.data import DataLoader

from ..data import SuperTileRNADataset, custom_collate_fn, filter_no_features, shard_rows
from ..train import predict
from ..vis import ViS
from .common import init_distributed, seed_everything


def main(argv=None):
    p = argparse.ArgumentParser(description='Getting features')
    p.add_argument('--ref_file', type=str, required=True)
    p.add_argument('--feature_path', type=str, default='')
    p.add_argument('--feature_use', type=str, default='cluster_features')
    p.add_argument('--folds', type=int, default=5)
    p.add_argument('--seed', type=int, default=99)
    p.add_argument('--batch_size', type=int, default=16)
    p.add_argument('--depth', type=int, default=6)
    p.add_argument('--num-heads', type=int, default=16)
    p.add_argument('--tcga_project', default='', type=str)
    p.add_argument('--save_dir', type=str, default='')
    p.add_argument('--exp_name', type=str, default='exp')
    p.add_argument('--model_dir', type=str, default=None)
    p.add_argument('--compute_dtype', default='fp32', choices=['fp32', 'bf16'])
    args = p.parse_args(argv)
    seed_everything(args.seed)
    rank, world, device = init_distributed()
    save_dir = os.path.join(args.save_dir, args.exp_name)
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    df = pd.read_csv(args.ref_file)
    df = filter_no_features(df, feature_path=args.feature_path, feature_name=args.feature_use)
    genes = [c[4:] for c in df.columns if "rna_" in c]
    if 'tcga_project' in df.columns and args.tcga_project:
        df = df[df['tcga_project'].isin([args.tcga_project])].reset_index(drop=True)
    lo, hi = shard_rows(df.shape[0], rank, world)
    test_dataset = SuperTileRNADataset(df.iloc[lo:hi].reset_index(drop=True), args.feature_path, args.feature_use)
    loader = DataLoader(test_dataset, num_workers=0, pin_memory=True, shuffle=False, batch_size=args.batch_size,
                        collate_fn=custom_collate_fn)
    cancer = args.tcga_project.split('-')[-1].lower()
    res_preds, res_random = [], []
    for fold in range(args.folds):
        name = f"sequoia-{cancer}-{fold}"
        src = os.path.join(args.model_dir, name) if args.model_dir else f"gevaertlab/{name}"
        model = ViS.from_pretrained(src, compute_dtype=args.compute_dtype)
        model.to(device)
        preds, wsis, projs = predict(model, loader)
        random_model = ViS(num_outputs=test_dataset.num_genes, input_dim=test_dataset.feature_dim, depth=args.depth,
                           nheads=args.num_heads, dimensions_f=64, dimensions_c=64, dimensions_s=64, device=str(device),
                           compute_dtype=args.compute_dtype).to(device)
        random_preds, _, _ = predict(random_model, loader)
        res_preds.append(preds)
        res_random.append(random_preds)
    avg_preds, avg_random = np.mean(res_preds, axis=0), np.mean(res_random, axis=0)
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, (avg_preds, avg_random, wsis))
        avg_preds = np.concatenate([g[0] for g in gathered])
        avg_random = np.concatenate([g[1] for g in gathered])
        wsis = np.concatenate([g[2] for g in gathered])
    if rank == 0:
        test_results = {'pred': pd.DataFrame(avg_preds, index=wsis, columns=genes),
                        'random': pd.DataFrame(avg_random, index=wsis, columns=genes)}
        with open(os.path.join(save_dir, 'test_results.pkl'), 'wb') as f:
            pickle.dump(test_results, f, protocol=pickle.HIGHEST_PROTOCOL)


if __name__ == '__main__':
    main()
