"""Worker of tests/test_gpu_multirank.py: one of N ranks that SHARE cuda:0 (a 1-GPU box), process group over gloo.

    multirank_worker.py spatial <ok_dir> <nx> <ny> <mode> <batch_windows> <stride> <holes 0|1> <head_chunk> <size small|full>
        ONE slide's windows and tiles dealt over the ranks (spatial.sliding_window_all_genes_sharded, BASELINE config 5's
        multi-GPU form) against the one-rank call in the same process: every rank's rows must be BIT-identical.
    multirank_worker.py cli <module> <args...>
        sequoia_pub_amd.cli.<module>.main(args) under the launcher's RANK / WORLD_SIZE (SQ_SHARE_GPU=1 -> cuda:0 + gloo)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import sequoia_pub_amd  # noqa: E402,F401


def bits(t):
    return t.contiguous().view(torch.int32)


def spatial(argv):
    import pandas as pd
    from sequoia_pub_amd import spatial as sp
    from sequoia_pub_amd.vis import ViS
    ok_dir, nx, ny, mode, bw, stride, holes, head_chunk, size = argv[0], int(argv[1]), int(argv[2]), argv[3], int(argv[4]), int(argv[5]), argv[6] == "1", int(argv[7]), argv[8]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    sp.HEAD_CHUNK = head_chunk
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    x, y = xs.ravel(), ys.ravel()
    if holes:
        keep = np.random.default_rng(5).random(x.size) < 0.85
        x, y = x[keep], y[keep]
    df = pd.DataFrame({"xcoord_tf": x, "ycoord_tf": y})
    cfg = dict(num_outputs=20820, input_dim=1024, depth=6, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64) if size == "full" \
        else dict(num_outputs=52, input_dim=128, depth=2, nheads=2, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    torch.manual_seed(31)                                  # the same model and the same slide on every rank
    m = ViS(**cfg, device="cuda:0", compute_dtype=mode).to("cuda:0").eval()
    feats = torch.randn(x.size, cfg["input_dim"], generator=torch.Generator().manual_seed(32)).cuda()

    out_l, ids, votes = sp.sliding_window_all_genes_sharded(x, y, feats, m, stride, batch_windows=bw, shard=(rank, world))
    out_1, votes_1 = sp.sliding_window_all_genes(x, y, feats, m, stride, batch_windows=bw)
    assert torch.equal(votes, votes_1)
    assert out_l.shape == (ids.numel(), cfg["num_outputs"]) and out_1.shape == (x.size, cfg["num_outputs"])
    same = torch.equal(bits(out_l), bits(out_1[ids]))      # bit patterns: NaN rows (tiles no kept window covers) compare equal, -0.0 != 0.0
    covered = int((votes[ids] > 0).sum())
    print(f"rank {rank}/{world}: {ids.numel()} of {x.size} tiles ({covered} covered), {int((votes > 0).sum())} covered in the slide, "
          f"windows/tile max {int(votes.max())}; rows bit-identical to the one-rank run: {same}", flush=True)
    assert same
    if covered:
        assert bool(torch.isfinite(out_l[votes[ids] > 0]).all())
    owned = [None] * world
    dist.all_gather_object(owned, ids.cpu().tolist())
    assert sorted(sum(owned, [])) == list(range(x.size)), "the ranks' tile sets must partition the slide"
    nb, slots = sp.window_batch_owner(int(sp.enumerate_windows(x, y, stride)[0].shape[0]), bw, world)
    if size == "full":
        assert nb >= 2 * world, "the full-size case must give every rank several window batches"

    genes = [3, 17, cfg["num_outputs"] - 1]
    d_s = sp.sliding_window_method(df, feats, m, genes, stride, batch_windows=bw, shard=(rank, world))
    d_1 = sp.sliding_window_method(df, feats, m, genes, stride, batch_windows=bw)
    for g in genes:
        assert list(d_s[g].keys()) == list(d_1[g].keys()) and len(d_1[g]) == int((votes > 0).sum())
        a = np.array(list(d_s[g].values()), dtype=np.float32)
        b = np.array(list(d_1[g].values()), dtype=np.float32)
        assert np.array_equal(a.view(np.int32), b.view(np.int32)), f"gene {g}: the gathered dictionary differs from the one-rank one"
    dist.barrier()
    open(os.path.join(ok_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def cli(argv):
    import importlib
    mod = importlib.import_module("sequoia_pub_amd.cli." + argv[0])
    mod.main(argv[1:])
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    {"spatial": spatial, "cli": cli}[sys.argv[1]](sys.argv[2:])
