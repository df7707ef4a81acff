"""Host logic (no GPU): feature store, dataset / collate / k-fold mirrors of src/read_data.py + src/utils.py,
slide sharding, and the world_size-2 (gloo) reductions the multi-GPU loops rely on."""
import os

import numpy as np
import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sequoia_pub_amd import store
from sequoia_pub_amd.data import (SuperTileRNADataset, custom_collate_fn, filter_no_features, patient_kfold, shard_rows)


def make_store(root, n=12, dim=64, genes=10, bad=()):
    rows = []
    rs = np.random.RandomState(0)
    for i in range(n):
        wsi = f"TCGA-XX-{i:04d}"
        d = os.path.join(root, "TCGA-BRCA", wsi)
        os.makedirs(d, exist_ok=True)
        f = store.File(os.path.join(d, wsi + ".h5"), "w")
        f.create_dataset("resnet_features", data=rs.randn(120, dim).astype(np.float32))
        if i not in bad:
            f.create_dataset("cluster_features", data=rs.randn(100, dim).astype(np.float32))
        f.close()
        rows.append(dict(wsi_file_name=wsi, patient_id=f"P{i // 2}", tcga_project="TCGA-BRCA",
                         **{f"rna_G{g}": float(rs.rand() * 6) for g in range(genes)}))
    return pd.DataFrame(rows)


def test_store_roundtrip_and_resume_guard(tmp_path):
    p = str(tmp_path / "a.h5")
    f = store.File(p, "w")
    f.create_dataset("resnet_features", data=np.arange(12, dtype=np.float32).reshape(3, 4))
    f.close()
    f = store.File(p, "r+")
    assert "cluster_features" not in f.keys()
    f.create_dataset("cluster_features", data=np.ones((2, 4), np.float32))
    with pytest.raises(Exception):
        f.create_dataset("cluster_features", data=np.ones((2, 4), np.float32))     # kmean_features.py:91-94 guard
    f.close()
    with store.File(p, "r") as g:
        assert sorted(g.keys()) == ["cluster_features", "resnet_features"]
        assert np.asarray(g["resnet_features"][:]).shape == (3, 4)
    with pytest.raises(Exception):
        store.File(str(tmp_path / "missing.h5"), "r")


def test_dataset_collate_filter(tmp_path):
    df = make_store(str(tmp_path), bad=(3,))
    kept = filter_no_features(df, str(tmp_path), "cluster_features")
    assert len(kept) == 11 and "TCGA-XX-0003" not in set(kept.wsi_file_name)
    ds = SuperTileRNADataset(df, str(tmp_path))
    assert ds.num_genes == 10 and ds.feature_dim == 64 and len(ds) == 12
    x, y, name, proj = ds[0]
    assert x.shape == (100, 64) and y.shape == (10,) and name == "TCGA-XX-0000" and proj == "TCGA-BRCA"
    assert ds[3][0] is None                                           # missing dataset -> None (read_data.py:51-54)
    batch = custom_collate_fn([ds[2], ds[3], ds[4]])
    assert batch[0].shape == (2, 100, 64) and list(batch[2]) == ["TCGA-XX-0002", "TCGA-XX-0004"]
    assert custom_collate_fn([ds[3]])[0] == []


def test_patient_kfold_matches_sklearn_protocol(tmp_path):
    df = make_store(str(tmp_path), n=40)
    tr, va, te = patient_kfold(df, n_splits=5)
    assert len(tr) == len(va) == len(te) == 5
    for a, b, c in zip(tr, va, te):
        pa, pb, pc = set(df.patient_id[a]), set(df.patient_id[b]), set(df.patient_id[c])
        assert not (pa & pb) and not (pa & pc) and not (pb & pc)       # patient-level split
        assert len(a) + len(b) + len(c) == len(df)
    assert sorted(np.concatenate(te).tolist()) == list(range(40))      # every slide is tested exactly once
    tr2, va2, te2 = patient_kfold(df, n_splits=5)
    assert all(np.array_equal(x, y) for x, y in zip(te, te2))          # random_state=0 -> reproducible


def test_shard_rows_cover_everything():
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(n, r, world) for r in range(world)]
            got = [i for lo, hi in spans for i in range(lo, hi)]
            assert got == list(range(n))


def _gloo_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sequoia_pub_amd.train import _all_mean
    # epoch means are means over ALL batches of ALL ranks (vit.py:182-184 semantics on the global batch list)
    vals = [1.0, 2.0, 3.0] if rank == 0 else [10.0]
    m = _all_mean(vals, torch.device("cpu"))
    assert abs(m - 4.0) < 1e-12
    # gradient exchange: sum of per-rank grads scaled by the GLOBAL element count == single-process gradient
    torch.manual_seed(0)
    w = torch.randn(5, 3)
    x = torch.randn(8, 3)
    y = torch.randn(8, 5)
    lo, hi = shard_rows(8, rank, world)
    pred = x[lo:hi] @ w.T
    g_local = (2.0 / (8 * 5)) * (pred - y[lo:hi]).T @ x[lo:hi]          # grad_scale = 2 / n_global
    dist.all_reduce(g_local)
    g_full = (2.0 / (8 * 5)) * ((x @ w.T) - y).T @ x
    assert torch.allclose(g_local, g_full, atol=1e-6)
    # sharded prediction gather (predict_independent_dataset counterpart)
    out = [None] * world
    dist.all_gather_object(out, list(range(lo, hi)))
    assert sum(out, []) == list(range(8))
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_patient_kfold_equals_reference_golden(golden_dir):
    """tests/golden/kfold.json: index lists produced by the reference's own patient_kfold (make_golden.py)."""
    import json
    for case in json.load(open(os.path.join(golden_dir, "kfold.json"))):
        df = pd.DataFrame(dict(patient_id=case["patient_id"]))
        tr, va, te = patient_kfold(df, n_splits=5, valid_size=case["valid_size"])
        for got, want in ((tr, case["train"]), (va, case["valid"]), (te, case["test"])):
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert g.tolist() == w


def test_checkpoint_policy_equals_reference_trace(golden_dir):
    """tests/golden/early_stop.json: which epochs the reference train() saved at and where it stopped, for scripted
    validation curves (make_golden.py gold_early_stop) -- the save / stop state machine must reproduce every row."""
    import json
    from sequoia_pub_amd.train import CheckpointPolicy
    cases = json.load(open(os.path.join(golden_dir, "early_stop.json")))
    assert len(cases) >= 32 and any(c["stopped_early"] for c in cases)
    for c in cases:
        pol = CheckpointPolicy(c["save_on"], c["stop_on"], c["patience"], c["delta"])
        saves, ran = [], 0
        for epoch, (loss, score) in enumerate(zip(c["losses"], c["scores"])):
            ran += 1
            if pol.observe(loss, score):
                saves.append(epoch)
            if pol.end_of_epoch(epoch) is not None:
                break
        assert saves == c["save_epochs"], (c["kind"], c["save_on"], c["stop_on"], c["patience"])
        assert ran == c["epochs_run"], (c["kind"], c["save_on"], c["stop_on"], c["patience"])


def test_numa_binding_helper_never_fails():
    """cli/common.bind_to_gpu_numa_node: parses sysfs cpulists and reports instead of raising when there is no GPU / no sysfs
    entry (this container); on an 8-GPU node it pins each rank's host threads next to its GPU."""
    from sequoia_pub_amd.cli.common import _parse_cpulist, bind_to_gpu_numa_node
    assert _parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and _parse_cpulist("") == set()
    import os
    import torch
    aff, nthr = os.sched_getaffinity(0), torch.get_num_threads()
    try:
        info = bind_to_gpu_numa_node(0)
        assert info["gpu"] == 0 and isinstance(info["bound"], bool) and (info["bound"] or "why" in info)
        os.environ["SQ_NO_NUMA_BIND"] = "1"
        off = bind_to_gpu_numa_node(0)
        assert off["bound"] is False and "SQ_NO_NUMA_BIND" in off["why"]
    finally:                                    # the helper changes process-wide state: leave the test process as it was
        os.environ.pop("SQ_NO_NUMA_BIND", None)
        os.sched_setaffinity(0, aff)
        torch.set_num_threads(nthr)


def test_bench_kernel_class_geometry():
    """bench.py maps a profiled kernel class to the (symbol, grid) rocprofv3 reports, for the per-class traffic table: the grids must
    follow the launchers' tilings (chains: 128-pixel tiles of 512 threads; tails: 64-pixel tiles of 256; halo 3x3: 256 x 128 of 512)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sq_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    k = b._kernel_of_class
    assert k("chainw_f16x3_c256_cn256_P196000") == ("chain_x3w_kernel<256, 256, true", 1532 * 512)
    assert k("chainw_f16x3_c128_cn256_P784000") == ("chain_x3w_kernel<128, 256, true", 6125 * 512)
    assert k("tail_f16x3_c64_cn64_ds_P3136000") == ("chain_x3_kernel<64, true, true, true", 49000 * 256)
    assert k("conv_f16x3_M196000_N256_K2304") == ("conv_halo_x3_kernel", 766 * 2 * 512)
    assert k("dual_f16x3_M196000_N1024_K256_K512")[1] == 1532 * 8 * 256
    assert k("gemm_f16x3_M196000_N256_K1024") == ("gemm_x3_kernel<256", 766 * 2 * 512)
    assert k("gemm_f32_M800_N2048_K2048_b1") is None
