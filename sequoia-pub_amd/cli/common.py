"""Shared CLI plumbing: seeds, device / torch.distributed setup (one process per GPU, RCCL)."""
import os
import random

import numpy as np
import torch


def seed_everything(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)


def init_distributed():
    """Returns (rank, world, device).  Under torchrun each rank owns LOCAL_RANK's GPU and the default
    process group is RCCL ('nccl' backend on ROCm); otherwise single GPU cuda:0."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local)
    if torch.cuda.is_available():
        torch.cuda.set_device(device)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    return rank, world, device


def ref_frame(path_csv, tcga_projects=None, start=None, end=None, dedupe=True):
    """compute_features_hdf5.py:72-85 / kmean_features.py:46-61: read, dedupe slides, filter, slice."""
    import pandas as pd
    df = pd.read_csv(path_csv)
    if dedupe:
        df = df.drop_duplicates(["wsi_file_name"])
    if tcga_projects:
        df = df[df['tcga_project'].isin(tcga_projects)]
    if start is not None and end is not None:
        df = df.iloc[start:end]
    elif start is not None:
        df = df.iloc[start:]
    elif end is not None:
        df = df.iloc[:end]
    return df
