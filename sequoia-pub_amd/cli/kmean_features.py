"""Cluster tokens -- counterpart of /root/reference/pre_processing/kmean_features.py (same flags; appends
dataset ``cluster_features`` [num_clusters, D] to the slide's feature file, skipping slides that have it or
have fewer patches than clusters).  KMeans + the per-label means run in ``sq_kmeans_fit``."""
import argparse
import os

import numpy as np
import torch

from .. import store
from ..data import shard_rows
from ..kmeans import kmeans_fit_batch
from .common import init_distributed, ref_frame, seed_everything


def main(argv=None):
    parser = argparse.ArgumentParser(description='Getting features')
    parser.add_argument('--ref_file', type=str, required=True)
    parser.add_argument('--patch_data_path', type=str, default=None)
    parser.add_argument('--feature_path', type=str, default="/examples/features")
    parser.add_argument('--num_clusters', type=int, default=100)
    parser.add_argument("--tcga_projects", default=None, type=str, nargs='*')
    parser.add_argument('--start', type=int, default=0)
    parser.add_argument('--end', type=int, default=None)
    parser.add_argument("--gtex", action="store_true")
    parser.add_argument('--gtex_tissue', type=str, default=None)
    parser.add_argument('--seed', type=int, default=99)
    parser.add_argument('--feat_name', type=str, default='resnet_features', help="dataset to cluster (the reference hard-codes 'resnet_features', :80)")
    args = parser.parse_args(argv)
    seed_everything(args.seed)
    rank, world, device = init_distributed()
    df = ref_frame(args.ref_file, args.tcga_projects, args.start, args.end)
    first_project = df.iloc[0]['tcga_project'] if len(df) else None
    lo, hi = shard_rows(df.shape[0], rank, world)
    df = df.iloc[lo:hi]
    for _, row in df.iterrows():
        WSI = row['wsi_file_name']
        if args.gtex:
            project = args.gtex_tissue
        else:
            project = first_project                      # kmean_features.py:70 uses the FIRST row's project
            WSI = WSI.replace('.svs', '')
        path = os.path.join(args.feature_path, project, WSI)
        try:
            f = store.File(os.path.join(path, WSI + '.h5'), "r+")
        except Exception:
            print(f'Cannot open file {path}')
            continue
        try:
            features = np.asarray(f[args.feat_name][:])
        except Exception:
            print(f'No resnet features for {path}')
            f.close()
            continue
        if features.shape[0] < args.num_clusters:
            print(f'{WSI} less number of patches than clusters')
            f.close()
            continue
        if 'cluster_features' in f.keys():
            print(f'{WSI}: Cluster feature already available')
            f.close()
            continue
        r = kmeans_fit_batch(torch.from_numpy(features).to(device), args.num_clusters, random_state=0)
        try:
            f.create_dataset("cluster_features", data=r["cluster_features"][0].cpu().numpy())
        except Exception as e:
            print(f"{WSI}: Error in creating cluster_feauture")
            print(e)
        f.close()
    print('Done!')


if __name__ == '__main__':
    main()
