"""Data feed -- host-side mirror of /root/reference/src/read_data.py:12-56 (``SuperTileRNADataset``) and
/root/reference/src/utils.py:10-41,79-110 (``custom_collate_fn``, ``filter_no_features``, ``patient_kfold``).
Pure host logic (pandas / numpy / sklearn splitters); the tensors it yields feed the HIP path."""
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from . import store


class SuperTileRNADataset(Dataset):
    """read_data.py:12-56.  ``feature_use`` names the dataset read per slide (the reference's ``__init__``
    reads an undefined ``self.feature_use`` and ``__getitem__`` hard-codes ``'cluster_features'``; the
    intended behaviour -- predict_independent_dataset.py:54 passes it as third argument -- is implemented)."""

    def __init__(self, csv_path, features_path, feature_use="cluster_features", quick=None):
        self.csv_path = csv_path
        self.quick = quick
        self.features_path = features_path
        self.feature_use = feature_use
        self.data = pd.read_csv(csv_path) if isinstance(csv_path, str) else csv_path
        self.rna_cols = [x for x in self.data.columns if 'rna_' in x]
        row = self.data.iloc[0]
        self.num_genes = len(self.rna_cols)
        with store.File(self._path(row), 'r') as f:
            self.feature_dim = f[self.feature_use][:].shape[1]
        # one vectorised conversion instead of a 20 823-column pandas slice per item
        self._rna = self.data[self.rna_cols].to_numpy(dtype=np.float32)

    def _path(self, row):
        path = os.path.join(self.features_path, row['tcga_project'], row['wsi_file_name'], row['wsi_file_name'] + '.h5')
        if 'GTEX' not in path:
            path = path.replace('.svs', '')        # read_data.py:45-46
        return path

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        row = self.data.iloc[idx]
        rna_data = torch.from_numpy(self._rna[idx].copy())
        path = self._path(row)
        try:
            with store.File(path, 'r') as f:
                features = torch.tensor(np.asarray(f[self.feature_use][:]), dtype=torch.float32)
        except Exception as e:                    # read_data.py:51-54: print, return None -> dropped by collate
            print(e)
            print(path)
            features = None
        return features, rna_data, row['wsi_file_name'], row['tcga_project']


def custom_collate_fn(batch):
    """utils.py:10-18: drop entries whose features failed to load, then default_collate (an all-bad batch
    collates to an empty list, which the loops skip: vit.py:159)."""
    batch = list(filter(lambda x: x[0] is not None, batch))
    if len(batch) == 0:
        return [], [], [], []
    return torch.utils.data.dataloader.default_collate(batch)


def filter_no_features(df, feature_path, feature_name):
    """utils.py:21-41."""
    print(f'Filtering WSIs that do not have {feature_name} features')
    projects = np.unique(df.tcga_project)
    all_wsis_with_features = []
    remove = []
    for proj in projects:
        wsis_with_features = os.listdir(os.path.join(feature_path, proj))
        for wsi in wsis_with_features:
            try:
                with store.File(os.path.join(feature_path, proj, wsi, wsi + '.h5'), "r") as f:
                    if feature_name not in list(f.keys()):
                        remove.append(wsi)
            except Exception:
                remove.append(wsi)
        all_wsis_with_features += wsis_with_features
    remove += df[~df['wsi_file_name'].isin(all_wsis_with_features)].wsi_file_name.values.tolist()
    print(f'Original shape: {df.shape}')
    df = df[~df['wsi_file_name'].isin(remove)].reset_index(drop=True)
    print(f'New shape: {df.shape}')
    return df


def patient_kfold(dataset, n_splits=5, random_state=0, valid_size=0.1):
    """utils.py:79-110: KFold(shuffle, random_state) over unique patient ids; 10 % of each train part -> val."""
    from sklearn.model_selection import KFold, train_test_split
    indices = np.arange(len(dataset))
    patients_unique = np.unique(dataset.patient_id)
    skf = KFold(n_splits, shuffle=True, random_state=random_state)
    train_idx, valid_idx, test_idx = [], [], []
    pid = np.array(dataset.patient_id)
    for ind_train, ind_test in skf.split(patients_unique):
        patients_train = patients_unique[ind_train]
        patients_test = patients_unique[ind_test]
        test_idx.append(indices[np.isin(pid, patients_test)])
        if valid_size > 0:
            patients_train, patients_valid = train_test_split(patients_train, test_size=valid_size, random_state=0)
            valid_idx.append(indices[np.isin(pid, patients_valid)])
        train_idx.append(indices[np.isin(pid, patients_train)])
    return train_idx, valid_idx, test_idx


def shard_rows(n_rows, rank, world):
    """Contiguous slide range of one rank -- the reference's --start/--end parallelisation
    (compute_features_hdf5.py:80-85, kmean_features.py:56-61) chosen automatically per GPU."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)
