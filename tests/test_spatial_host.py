"""Host-side window bookkeeping of the sliding-window path (spatial.py), CPU only: the device-side enumeration and the sync-free
vote lists against the numpy restatement of visualize.py:35-102's loop order."""
import numpy as np
import pytest
import torch

import sequoia_pub_amd  # noqa: F401
from sequoia_pub_amd.spatial import enumerate_windows, enumerate_windows_device, tile_window_lists


@pytest.mark.parametrize("stride", [1, 3, 10])
@pytest.mark.parametrize("holes,shuffled", [(False, False), (True, False), (True, True)])
def test_device_enumeration_and_bounded_vote_lists_equal_the_numpy_form(stride, holes, shuffled):
    rng = np.random.default_rng(7 + stride)
    xs, ys = np.meshgrid(np.arange(37), np.arange(29), indexing="ij")
    x, y = xs.ravel(), ys.ravel()
    if holes:
        keep = rng.random(x.size) < 0.8
        x, y = x[keep], y[keep]
    if shuffled:                                   # df order need not follow the grid: members are sorted by df position
        perm = rng.permutation(x.size)
        x, y = x[perm], y[perm]
    m, _ = enumerate_windows(x, y, stride)
    md = enumerate_windows_device(x, y, stride, "cpu")
    assert m.shape == tuple(md.shape) and np.array_equal(m, md.numpy())
    l0, c0 = tile_window_lists(m, x.size, "cpu")
    l1, c1 = tile_window_lists(md, x.size, "cpu", max_votes=(-(-10 // stride)) ** 2)
    V = l0.shape[1]
    assert torch.equal(c0, c1) and torch.equal(l0, l1[:, :V]) and bool((l1[:, V:] == -1).all())


def test_device_enumeration_rejects_duplicates_and_handles_empty_grids():
    with pytest.raises(ValueError):
        enumerate_windows_device(np.array([0, 1, 1]), np.array([0, 2, 2]), 1, "cpu")
    assert enumerate_windows_device(np.array([0]), np.array([0]), 1, "cpu").shape == (0, 100)
    assert enumerate_windows_device(np.arange(5), np.arange(5), 1, "cpu").shape == (0, 100)      # windows exist, none holds > 50 tiles


def _gather_worker(rank, world, port, W, B):
    """One of `world` gloo ranks: fills its window-batch slots (batch b -> rank b % world, slot b // world) with the window ids,
    all-gathers exactly as spatial.sliding_window_all_genes_sharded does, and checks that the remapped vote lists address every
    window's own row."""
    import os
    import torch.distributed as dist
    from sequoia_pub_amd.spatial import gathered_row_of_window, window_batch_owner
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nb, slots = window_batch_owner(W, B, world)
    local = torch.full((slots * B, 3), float("nan"))
    for b in range(rank, nb, world):
        s, n = b * B, min(B, W - b * B)
        o = (b // world) * B
        local[o:o + n] = torch.arange(s, s + n, dtype=torch.float32).unsqueeze(1).expand(n, 3)
    allv = torch.empty(world * slots * B, 3)
    dist.all_gather(list(allv.chunk(world)), local)
    w = torch.cat([torch.arange(W), torch.tensor([-1, -1])])
    rows = gathered_row_of_window(w, B, world, slots)
    assert bool((rows[W:] == -1).all()) and rows[:W].unique().numel() == W
    assert torch.equal(allv[rows[:W], 0], torch.arange(W, dtype=torch.float32))
    assert gathered_row_of_window(W - 1, B, world, slots) == int(rows[W - 1]) and gathered_row_of_window(-1, B, world, slots) == -1
    dist.destroy_process_group()


@pytest.mark.parametrize("W,B", [(47, 8), (16, 8), (5, 8), (33, 4)])
def test_window_batches_dealt_over_two_gloo_ranks_gather_back_to_every_windows_row(W, B):
    """The N > 1 bookkeeping of BASELINE config 5's window sharding on world_size 2 over gloo (CPU): ragged last batch, a rank
    with an empty last slot (16 / 8 -> one batch each; 5 / 8 -> rank 1 owns nothing)."""
    import os
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + W * 7 + B) % 2000
    mp.spawn(_gather_worker, args=(2, port, W, B), nprocs=2, join=True)


def test_one_rank_gather_map_is_the_identity_and_shard_arguments_are_checked():
    from sequoia_pub_amd.spatial import _shard_info, gathered_row_of_window, max_votes_per_tile, window_batch_owner
    nb, slots = window_batch_owner(47769, 1024, 1)
    assert (nb, slots) == (47, 47)
    w = torch.arange(47769)
    assert torch.equal(gathered_row_of_window(w, 1024, 1, slots), w)
    assert window_batch_owner(47769, 1024, 8) == (47, 6)
    assert [max_votes_per_tile(s) for s in (1, 3, 5, 10)] == [100, 16, 4, 1]
    assert _shard_info(None) == (0, 1, None)
    with pytest.raises(ValueError):
        _shard_info((2, 2))
    with pytest.raises(RuntimeError):
        _shard_info((0, 2))                        # no process group in this process


class _FakeExtractor:
    """Stands in for the patch embedder on the CPU: a deterministic function of the tile's pixels, 1024 columns."""
    def extract_patches_u8(self, t):
        f = t.reshape(t.shape[0], -1).float()
        return torch.stack([f.mean(1), f.std(1), f[:, 0], f[:, -1]], 1).repeat(1, 256)


def _embed_worker(rank, world, port, n_tiles, chunk, out_dir):
    import os
    import pandas as pd
    import torch.distributed as dist
    from sequoia_pub_amd.cli.visualize import embed_tiles
    from sequoia_pub_amd.patchgen import ArraySlide
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(2)
    slide = ArraySlide([rs.randint(0, 256, (64, 16 * n_tiles + 16, 3), dtype=np.uint8)])
    df = pd.DataFrame({"xcoord": np.arange(n_tiles) * 16, "ycoord": np.zeros(n_tiles, dtype=int)})
    one = embed_tiles(slide, df, 16, 16, _FakeExtractor(), "cpu", chunk=chunk)
    both = embed_tiles(slide, df, 16, 16, _FakeExtractor(), "cpu", chunk=chunk, shard=(rank, world))
    assert both.shape == one.shape == (n_tiles, 1024) and torch.equal(both, one), "the gathered cache differs from the one-rank cache"
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("n_tiles,chunk", [(23, 4), (8, 4), (3, 4), (0, 4)])
def test_tile_cache_embedded_over_two_gloo_ranks_equals_the_one_rank_cache(tmp_path, n_tiles, chunk):
    """cli/visualize.embed_tiles under two ranks (CPU, gloo, a stand-in extractor): tile chunks dealt round-robin, one all-gather,
    rows back in df order -- ragged last chunk, one chunk per rank, a rank without a chunk, a slide without valid tiles."""
    import os
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + 31 * n_tiles + chunk) % 2000
    mp.spawn(_embed_worker, args=(2, port, n_tiles, chunk, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").read_text() == "ok" for r in range(2))
