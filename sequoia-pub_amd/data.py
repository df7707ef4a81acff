"""Data feed -- host-side counterpart of /root/reference/src/read_data.py:12-56 (``SuperTileRNADataset``) and
/root/reference/src/utils.py:10-41,79-110 (``custom_collate_fn``, ``filter_no_features``, ``patient_kfold``).
Pure host logic (pandas / numpy / sklearn splitters); the tensors it yields feed the HIP path.

Same inputs, outputs and file rules as the reference; organised around whole-frame passes: store paths, names and
targets are materialised once per dataset, the feature filter scans each project directory once into a
name -> usable map, and the k-fold split is a patient -> role table expanded to row indices."""
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from . import store


def slide_store_path(features_path, project, wsi_file_name):
    """``<features_path>/<project>/<WSI>/<WSI>.h5``; TCGA names lose their ``.svs`` suffix, GTEx paths keep it
    (read_data.py:45-46)."""
    path = os.path.join(features_path, project, wsi_file_name, wsi_file_name + '.h5')
    return path if 'GTEX' in path else path.replace('.svs', '')


class SuperTileRNADataset(Dataset):
    """One item per reference-CSV row: (features f32 [100, D] or None, rna f32 [G], wsi_file_name, tcga_project)
    -- read_data.py:12-56.  ``feature_use`` names the dataset read per slide (the reference's ``__init__`` reads an
    undefined ``self.feature_use`` and ``__getitem__`` hard-codes ``'cluster_features'``; the intended behaviour
    -- predict_independent_dataset.py:54 passes it as third argument -- is implemented)."""

    def __init__(self, csv_path, features_path, feature_use="cluster_features", quick=None):
        self.csv_path = csv_path
        self.quick = quick
        self.features_path = features_path
        self.feature_use = feature_use
        self.data = pd.read_csv(csv_path) if isinstance(csv_path, str) else csv_path
        self.rna_cols = [c for c in self.data.columns if 'rna_' in c]
        self.num_genes = len(self.rna_cols)
        # whole-frame materialisation: no 20 823-column pandas row slice per item
        self._targets = self.data[self.rna_cols].to_numpy(dtype=np.float32)
        self._names = self.data['wsi_file_name'].to_numpy()
        self._projects = self.data['tcga_project'].to_numpy()
        self._paths = [slide_store_path(features_path, p, w) for p, w in zip(self._projects, self._names)]
        with store.File(self._paths[0], 'r') as f:
            self.feature_dim = f[self.feature_use][:].shape[1]

    def __len__(self):
        return len(self._paths)

    def _load(self, path):
        """The slide's token matrix, or None (with the reference's two prints) when the store cannot give it:
        such rows are dropped by the collate function (read_data.py:51-54)."""
        try:
            with store.File(path, 'r') as f:
                return torch.tensor(np.asarray(f[self.feature_use][:]), dtype=torch.float32)
        except Exception as e:
            print(e)
            print(path)
            return None

    def __getitem__(self, idx):
        return (self._load(self._paths[idx]), torch.from_numpy(self._targets[idx].copy()),
                self._names[idx], self._projects[idx])


def custom_collate_fn(batch):
    """utils.py:10-18: drop entries whose features failed to load, then default_collate (an all-bad batch
    collates to an empty list, which the loops skip: vit.py:159)."""
    usable = [item for item in batch if item[0] is not None]
    if not usable:
        return [], [], [], []
    return torch.utils.data.dataloader.default_collate(usable)


def _has_dataset(path, name):
    try:
        with store.File(path, "r") as f:
            return name in f.keys()
    except Exception:
        return False


def filter_no_features(df, feature_path, feature_name):
    """Rows of df whose slide has a readable store holding `feature_name` (utils.py:21-41).  A slide name that
    is unusable under ANY of the frame's projects is dropped everywhere, like the reference's remove-by-name."""
    print(f'Filtering WSIs that do not have {feature_name} features')
    usable = {}
    for proj in np.unique(df.tcga_project):
        for wsi in os.listdir(os.path.join(feature_path, proj)):
            ok = _has_dataset(os.path.join(feature_path, proj, wsi, wsi + '.h5'), feature_name)
            usable[wsi] = usable.get(wsi, True) and ok
    keep = df['wsi_file_name'].map(lambda w: usable.get(w, False)).to_numpy(dtype=bool)
    print(f'Original shape: {df.shape}')
    df = df[keep].reset_index(drop=True)
    print(f'New shape: {df.shape}')
    return df


def patient_kfold(dataset, n_splits=5, random_state=0, valid_size=0.1):
    """Patient-level cross-validation (utils.py:79-110): KFold(shuffle, random_state) over the sorted unique patient
    ids; `valid_size` of each fold's training patients (train_test_split, random_state=0) becomes the validation
    part.  Returns (train_idx, valid_idx, test_idx): per fold, ascending row indices."""
    from sklearn.model_selection import KFold, train_test_split
    patients, patient_of_row = np.unique(np.asarray(dataset.patient_id), return_inverse=True)
    TRAIN, VALID, TEST = 0, 1, 2
    folds = {TRAIN: [], VALID: [], TEST: []}
    for keep_pos, test_pos in KFold(n_splits, shuffle=True, random_state=random_state).split(patients):
        role = np.full(len(patients), TRAIN, dtype=np.int8)
        role[test_pos] = TEST
        if valid_size > 0:
            _, valid_pos = train_test_split(keep_pos, test_size=valid_size, random_state=0)
            role[valid_pos] = VALID
        row_role = role[patient_of_row]
        folds[TEST].append(np.flatnonzero(row_role == TEST))
        folds[TRAIN].append(np.flatnonzero(row_role == TRAIN))
        if valid_size > 0:
            folds[VALID].append(np.flatnonzero(row_role == VALID))
    return folds[TRAIN], folds[VALID], folds[TEST]


def shard_rows(n_rows, rank, world):
    """Contiguous slide range of one rank -- the reference's --start/--end parallelisation
    (compute_features_hdf5.py:80-85, kmean_features.py:56-61) chosen automatically per GPU."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)
