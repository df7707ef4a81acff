"""Pre-training on a whole cohort without validation split -- counterpart of /root/reference/src/pretrain_gtex.py
(same flags; `train(..., phases=['train'])`, AdamW lr 3e-3, checkpoint `model_best.pt` under
`<save_dir>/<date>_<exp_name>/`).  `--model he2rna` builds the MLP comparator of pretrain_gtex.py:102-105 and trains it
with he2rna.fit (:118-120: Adam lr 3e-3, no validation, whole model pickled as `model.pt`)."""
import argparse
import datetime
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import DataLoader

from ..data import SuperTileRNADataset, custom_collate_fn, filter_no_features
from ..train import train
from ..vis import ViS
from .common import seed_everything


def main(argv=None):
    p = argparse.ArgumentParser(description='Getting features')
    p.add_argument('--save_dir', type=str, default="/examples/pretrained_model", help='save directory')
    p.add_argument('--path_csv', type=str, default="/examples/ref_file.csv", help='path to reference file with gene expression data')
    p.add_argument('--feature_path', type=str, default="/examples/features", help='path to resnet and clustered features')
    p.add_argument('--exp_name', type=str, default="exp", help='Experiment name used to create saved model name')
    p.add_argument('--log', type=int, default=0, help='whether to log the loss')
    p.add_argument('--model', type=str, default='vis', help='model to pretrain, "he2rna" for MLP aggregation, "vit" for transformer aggregation or "vis" for linearized transformer aggregation')
    p.add_argument('--seed', type=int, default=99)
    p.add_argument('--num_epochs', type=int, default=200)
    p.add_argument('--batch_size', type=int, default=16)
    p.add_argument('--n_workers', type=int, default=8)
    p.add_argument('--checkpoint', type=str, default=None)
    p.add_argument('--quick', type=int, default=0, help='Whether to run a quick exp for debugging')
    p.add_argument('--compute_dtype', default='fp32', choices=['fp32', 'bf16'])
    args = p.parse_args(argv)

    seed_everything(args.seed)
    date = '{date:%Y-%m-%d}'.format(date=datetime.datetime.now())
    args.exp_name = date if args.exp_name == "" else date + "_" + args.exp_name        # pretrain_gtex.py:66-69
    save_dir = os.path.join(args.save_dir, args.exp_name)
    os.makedirs(save_dir, exist_ok=True)
    run = None
    if args.log:
        try:
            import wandb
            run = wandb.init(project="sequoia", config=args, name=args.exp_name)
        except Exception:
            print('wandb not available: logging to stdout only')

    device = torch.device("cuda:0")
    df = pd.read_csv(args.path_csv)
    df = filter_no_features(df, feature_path=args.feature_path, feature_name='cluster_features')
    if args.quick:
        df = df.iloc[0:20, :]
        args.num_epochs = 5
    dataset = SuperTileRNADataset(df, args.feature_path)
    # the dataset reads small per-slide files; worker processes would each need the library's device context
    dataloader = DataLoader(dataset, num_workers=0, pin_memory=True, shuffle=True, batch_size=args.batch_size,
                            collate_fn=custom_collate_fn)
    if args.model == 'vis':
        model = ViS(num_outputs=dataset.num_genes, input_dim=dataset.feature_dim, depth=6, nheads=16, dimensions_f=64,
                    dimensions_c=64, dimensions_s=64, device=str(device), compute_dtype=args.compute_dtype)
    elif args.model == 'vit':
        from ..vit import ViT
        model = ViT(num_outputs=dataset.num_genes, dim=dataset.feature_dim, depth=6, heads=16, mlp_dim=2048, dim_head=64,
                    device=str(device), compute_dtype=args.compute_dtype)
    elif args.model == 'he2rna':
        from ..he2rna import HE2RNA, fit
        model = HE2RNA(input_dim=dataset.feature_dim, layers=[256, 256], ks=[1, 2, 5, 10, 20, 50, 100],
                       output_dim=dataset.num_genes, device=str(device))
    else:
        raise SystemExit('please specify correct model name, "vit" or "he2rna"')
    if args.checkpoint is not None:
        model.load_state_dict(torch.load(args.checkpoint, map_location='cpu'))
    model = model.to(device)
    if args.model == 'he2rna':
        # pretrain_gtex.py:118-120 (200 epochs by default; --quick caps the run the way it does for the other models)
        model = fit(model=model, lr=3e-3, train_loader=dataloader, valid_loader=None, test_loader=None,
                    params={'max_epochs': args.num_epochs} if args.quick else {}, fold=None, optimizer=None, path=save_dir)
        print('Finished pre-training')
        return model, save_dir
    model = train(model, {'train': dataloader}, None, num_epochs=args.num_epochs, phases=['train'], save_dir=save_dir, run=run,
                  lr=3e-3)
    print('Finished pre-training')
    return model, save_dir


if __name__ == '__main__':
    main()
