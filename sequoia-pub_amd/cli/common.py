"""Shared CLI plumbing: seeds, device / torch.distributed setup (one process per GPU, RCCL)."""
import os
import random

import numpy as np
import torch


def seed_everything(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa_node(device_index):
    """Pin this process's launch (main) thread to the CPUs of the NUMA node its GPU hangs off (PCI address -> sysfs numa_node ->
    node cpulist -> sched_setaffinity).  On an 8-GPU node every rank then drives its GPU, and first-touches its pinned
    staging buffers, from the local socket instead of wherever the launcher started it; torch's intra-op pool is capped at 16
    threads with it (see below).  Never fails: returns a dict that
    says what was done (bench.py prints it), or why nothing was."""
    info = {"gpu": int(device_index), "node": None, "cpus": None, "bound": False}
    if os.environ.get("SQ_NO_NUMA_BIND", "0") not in ("", "0"):
        info["why"] = "SQ_NO_NUMA_BIND is set"
        return info
    try:
        import torch
        prop = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
        info["pci"] = bdf
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        info["node"] = node
        if node < 0:
            info["why"] = "the platform reports no NUMA affinity for this device (single node or virtualised)"
            return info
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        if not cpus:
            info["why"] = "none of the node's CPUs is in this process's allowed set"
            return info
        os.sched_setaffinity(0, cpus)           # (pid 0 = the calling thread; threads it starts later inherit the mask)
        info["cpus"] = len(cpus)
        info["bound"] = True
        # Host-side glue (BN folding, plane splitting, index lists) is a stream of small torch ops; with the default pool of one thread
        # per machine core they crawl once the launch thread is confined to a node -- spinning workers fight it for its CPUs: measured
        # on a 2 x 128-CPU box, folding + splitting a ResNet-50: 23 s bound / 0.9 s unbound with the default pool, 0.1 s with 16
        # threads either way (bench.py: 6.5 -> ~3 minutes).  Callers that time CPU work set their own thread count afterwards.
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)
            info["torch_threads"] = 16
    except Exception as e:                      # sysfs layout, permissions, attribute names: never worth failing a run for
        info["why"] = f"{type(e).__name__}: {e}"
    return info


def init_distributed():
    """Returns (rank, world, device).  Under torchrun each rank owns LOCAL_RANK's GPU and the default
    process group is RCCL ('nccl' backend on ROCm); otherwise single GPU cuda:0.  SQ_SHARE_GPU=1 (debugging aid for 1-GPU
    boxes, the CLI twin of bench.py's SQ_BENCH_SHARE_GPU): every rank uses cuda:0 and the collectives go through gloo --
    exercises the multi-process control flow, not RCCL."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("SQ_SHARE_GPU") == "1"
    device = torch.device("cuda", 0 if share else local)
    if torch.cuda.is_available():
        if world > 1 and not share and torch.cuda.device_count() <= local:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local} needs {local + 1} visible GPUs, this box has {torch.cuda.device_count()} "
                             "(one rank per GPU over RCCL; SQ_SHARE_GPU=1 runs the ranks on cuda:0 over gloo to exercise the control flow only)")
        torch.cuda.set_device(device)
        if world > 1:
            # narrows this thread's CPU mask (threads / DataLoader workers started later inherit it) and caps torch's intra-op
            # pool at 16 threads; SQ_NO_NUMA_BIND=1 opts out
            info = bind_to_gpu_numa_node(device.index)
            if rank == 0:
                print(f"[sequoia-pub_amd] host NUMA binding: {info}", flush=True)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl" if torch.cuda.is_available() and not share else "gloo")
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
