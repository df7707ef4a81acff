"""Training / k-fold evaluation -- counterpart of /root/reference/src/main.py (same flags and outputs:
``{train,val,test}_{i}.npy``, ``model_best[_i].pt``, ``test_results.pkl``).  The shipped reference file does
not parse (main.py:93); this implements what it intends.  Under torchrun the training batch is
sharded across ranks (DDP: RCCL all-reduce of the flat gradient), evaluation runs on every rank identically."""
import argparse
import os
import pickle

import numpy as np
import pandas as pd
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from ..data import SuperTileRNADataset, custom_collate_fn, filter_no_features, patient_kfold
from ..train import evaluate, train
from ..vis import ViS
from .common import init_distributed, seed_everything


def build_model(args, num_outputs, feature_dim, device):
    if args.model_type == 'vit':                       # src/main.py:141-143,160-163
        from ..vit import ViT
        return ViT(num_outputs=num_outputs, dim=feature_dim, depth=args.depth, heads=args.num_heads, mlp_dim=2048,
                   dim_head=64, device=str(device), compute_dtype=args.compute_dtype)
    if args.model_type != 'vis':
        raise SystemExit('please specify correct model type "vit" or "vis"')
    return ViS(num_outputs=num_outputs, input_dim=feature_dim, depth=args.depth, nheads=args.num_heads,
               dimensions_f=64, dimensions_c=64, dimensions_s=64, device=str(device), compute_dtype=args.compute_dtype)


def main(argv=None):
    p = argparse.ArgumentParser(description='Getting features')
    p.add_argument('--src_path', type=str, default='')
    p.add_argument('--ref_file', type=str, default=None)
    p.add_argument('--sample-percent', type=float, default=None)
    p.add_argument('--tcga_projects', default=None, type=str)
    p.add_argument('--feature_path', type=str, default="features/")
    p.add_argument('--save_dir', type=str, default='saved_exp')
    p.add_argument('--cohort', type=str, default="TCGA")
    p.add_argument('--exp_name', type=str, default="exp")
    p.add_argument('--filter_no_features', type=int, default=1)
    p.add_argument('--log', type=str)
    p.add_argument('--model_type', type=str, default='vis')
    p.add_argument('--depth', type=int, default=6)
    p.add_argument('--num-heads', type=int, default=16)
    p.add_argument('--seed', type=int, default=99)
    p.add_argument('--lr', type=float, default=1e-3)
    p.add_argument('--batch_size', type=int, default=16)
    p.add_argument('--checkpoint', type=str, default=None)
    p.add_argument('--train', action="store_true")
    p.add_argument('--num_epochs', type=int, default=200)
    p.add_argument('--change_num_genes', type=int, default=0)
    p.add_argument('--num_genes', type=int, default=None)
    p.add_argument('--k', type=int, default=5)
    p.add_argument('--save_on', type=str, default='loss')
    p.add_argument('--stop_on', type=str, default='loss')
    p.add_argument('--compute_dtype', default='fp32', choices=['fp32', 'bf16'])
    p.add_argument('--grad_exchange', default='fp32', choices=['fp32', 'bf16'],
                   help='wire format of the gradient all-reduce under torchrun (bf16: half the bytes, gradients rounded to bf16 on the wire)')
    args = p.parse_args(argv)

    seed_everything(args.seed)
    rank, world, device = init_distributed()
    g = torch.Generator()
    g.manual_seed(0)
    save_dir = os.path.join(args.src_path, args.save_dir, args.cohort, args.exp_name)
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    run = None
    if args.log and rank == 0:
        try:
            import wandb
            run = wandb.init(project=args.log, config=args, name=args.exp_name)
        except Exception:
            print('wandb not available: logging to stdout only')

    df = pd.read_csv(args.ref_file)
    if args.sample_percent is not None:
        df = df.sample(frac=args.sample_percent).reset_index(drop=True)
    if ('tcga_project' in df.columns) and args.tcga_projects:
        df = df[df['tcga_project'].isin(args.tcga_projects.split(','))].reset_index(drop=True)
    if args.filter_no_features:
        df = filter_no_features(df, feature_path=args.feature_path, feature_name='cluster_features')

    train_idxs, val_idxs, test_idxs = patient_kfold(df, n_splits=args.k)
    test_results_splits = {}
    for i, (train_idx, val_idx, test_idx) in enumerate(zip(train_idxs, val_idxs, test_idxs)):
        train_df, val_df, test_df = df.iloc[train_idx], df.iloc[val_idx], df.iloc[test_idx]
        if rank == 0:
            np.save(save_dir + '/train_' + str(i) + '.npy', np.unique(train_df.patient_id))
            np.save(save_dir + '/val_' + str(i) + '.npy', np.unique(val_df.patient_id))
            np.save(save_dir + '/test_' + str(i) + '.npy', np.unique(test_df.patient_id))
        train_dataset = SuperTileRNADataset(train_df, args.feature_path)
        val_dataset = SuperTileRNADataset(val_df, args.feature_path)
        test_dataset = SuperTileRNADataset(test_df, args.feature_path)
        num_outputs, feature_dim = train_dataset.num_genes, train_dataset.feature_dim
        sampler = DistributedSampler(train_dataset, world, rank, shuffle=True, seed=0) if world > 1 else None
        train_dataloader = DataLoader(train_dataset, num_workers=0, pin_memory=True, shuffle=sampler is None, sampler=sampler,
                                      batch_size=args.batch_size, collate_fn=custom_collate_fn, generator=g)
        val_dataloader = DataLoader(val_dataset, num_workers=0, pin_memory=True, shuffle=True, batch_size=args.batch_size,
                                    collate_fn=custom_collate_fn)
        test_dataloader = DataLoader(test_dataset, num_workers=0, pin_memory=True, shuffle=False, batch_size=args.batch_size,
                                     collate_fn=custom_collate_fn)

        if args.checkpoint and args.change_num_genes:                        # fine-tune from another gene set (:138-157)
            model = build_model(args, args.change_num_genes, feature_dim, device)
            model.load_state_dict(torch.load(os.path.join(args.checkpoint), map_location='cpu'))
            print(f'Loaded model from {args.checkpoint}')
            model.linear_head = torch.nn.Sequential(torch.nn.LayerNorm(feature_dim), torch.nn.Linear(feature_dim, num_outputs))
        else:
            model = build_model(args, num_outputs, feature_dim, device)
        if args.checkpoint and not args.change_num_genes:
            suff = f'_{i}' if i > 0 else ''
            model_path = args.checkpoint + f'model_best{suff}.pt'
            print(f'Loading model from {model_path}')
            model.load_state_dict(torch.load(model_path, map_location='cpu'))
        model.to(device)
        if world > 1:                                                        # same initial weights on every rank
            torch.distributed.broadcast(model.flat.data, src=0)
        dataloaders = {'train': train_dataloader, 'val': val_dataloader}
        if args.train:
            model = train(model, dataloaders, None, num_epochs=args.num_epochs, run=run, split=i, save_on=args.save_on,
                          stop_on=args.stop_on, delta=0.5, save_dir=save_dir, lr=args.lr, grad_exchange=args.grad_exchange)
        preds, real, wsis, projs = evaluate(model, test_dataloader, run=run, suff='_' + str(i), verbose=rank == 0)
        random_model = build_model(args, num_outputs, feature_dim, device).to(device)
        random_preds, _, _, _ = evaluate(random_model, test_dataloader, run=run, suff='_' + str(i) + '_rand', verbose=False)
        test_results_splits[f'split_{i}'] = {'real': real, 'preds': preds, 'random': random_preds,
                                             'wsi_file_name': wsis, 'tcga_project': projs}
    test_results_splits['genes'] = [x[4:] for x in df.columns if 'rna_' in x]
    if rank == 0:
        with open(os.path.join(save_dir, 'test_results.pkl'), 'wb') as f:
            pickle.dump(test_results_splits, f, protocol=pickle.HIGHEST_PROTOCOL)


if __name__ == '__main__':
    main()
