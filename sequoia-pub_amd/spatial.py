"""Sliding-window spatial inference -- host-side mirror of /root/reference/spatial_vis/visualize.py:35-102
(``sliding_window_method``): 10x10-tile windows over the tile grid at a given stride, kept if they hold more
than 50 tiles, zero-padded to 100 tokens, one model forward per window, the window's prediction written to (or
averaged over, for stride < 10) every tile of the window.

Differences in HOW (not what): every tile is embedded ONCE into a feature cache (the reference re-embeds a tile
in every window that contains it, up to 100 times); windows are enumerated with tensor ops and run through the
model in large batches (BASELINE config 5: ~47k windows of 100 tokens) that are gathered from the cache on the fly;
the linear head is applied once per tile, after the vote.  ``literal_2d=True`` reproduces the
reference's 2-D input quirk (the prediction depends only on the window's first tile)."""
import os

import numpy as np
import torch

from . import _lib


def enumerate_windows(xtf, ytf, stride, size=10, min_tiles=50):
    """xtf, ytf: integer grid coordinates per tile (df order).  Returns (members [W, size*size] int64 with -1
    padding, tiles of a window in ascending df position as the reference's boolean mask yields them,
    origins [W, 2])."""
    xtf = np.asarray(xtf, dtype=np.int64)
    ytf = np.asarray(ytf, dtype=np.int64)
    max_x, max_y = int(xtf.max()), int(ytf.max())
    grid = np.full((max_x + size + 1, max_y + size + 1), -1, dtype=np.int64)
    # a later duplicate coordinate would overwrite an earlier one; the reference keeps both in the window --
    # real tile grids have unique coordinates, duplicates are rejected
    if len(set(zip(xtf.tolist(), ytf.tolist()))) != len(xtf):
        raise ValueError("duplicate tile coordinates")
    grid[xtf, ytf] = np.arange(len(xtf))
    xs = np.arange(0, max_x, stride)
    ys = np.arange(0, max_y, stride)
    if len(xs) == 0 or len(ys) == 0:
        return np.zeros((0, size * size), np.int64), np.zeros((0, 2), np.int64)
    win = np.lib.stride_tricks.sliding_window_view(grid, (size, size))[xs][:, ys]          # [nx, ny, size, size]
    members = win.reshape(len(xs) * len(ys), size * size)
    key = np.where(members >= 0, members, np.iinfo(np.int64).max)
    members = np.take_along_axis(members, np.argsort(key, axis=1, kind="stable"), axis=1)    # valid ascending, -1 last
    keep = (members >= 0).sum(1) > min_tiles
    origins = np.stack(np.meshgrid(xs, ys, indexing="ij"), -1).reshape(-1, 2)
    return members[keep], origins[keep]


def enumerate_windows_device(xtf, ytf, stride, device, size=10, min_tiles=50):
    """enumerate_windows with the [W, size*size] member table built ON THE DEVICE and nothing the host waits for: the host
    decides which windows are kept from an integral image of the occupancy grid (a few hundred microseconds), the device
    unfolds the index grid, sorts every window's members (valid ascending, -1 last) and keeps those rows.  At BASELINE config 5's
    size the numpy form is 50-140 ms of host time per slide during which the GPU idles.  Returns int64 [W, size*size] on `device`."""
    xtf = np.asarray(xtf, dtype=np.int64)
    ytf = np.asarray(ytf, dtype=np.int64)
    max_x, max_y = int(xtf.max()), int(ytf.max())
    if np.unique(xtf * (max_y + 1) + ytf).size != xtf.size:
        raise ValueError("duplicate tile coordinates")
    xs = np.arange(0, max_x, stride)
    ys = np.arange(0, max_y, stride)
    if len(xs) == 0 or len(ys) == 0:
        return torch.zeros((0, size * size), dtype=torch.int64, device=device)
    gx, gy = max_x + size + 1, max_y + size + 1
    occ = np.zeros((gx + 1, gy + 1), dtype=np.int64)
    occ[xtf + 1, ytf + 1] = 1
    ii = occ.cumsum(0).cumsum(1)                                   # ii[a, b] = tiles with x < a and y < b
    X, Y = xs[:, None], ys[None, :]
    n_in = ii[X + size, Y + size] - ii[X, Y + size] - ii[X + size, Y] + ii[X, Y]
    keep_idx = torch.from_numpy(np.flatnonzero((n_in > min_tiles).ravel())).to(device, non_blocking=True)
    xy = torch.from_numpy(np.stack([xtf, ytf])).to(device, non_blocking=True)
    grid = torch.full((gx, gy), -1, dtype=torch.int64, device=device)
    grid[xy[0], xy[1]] = torch.arange(xtf.size, device=device)
    win = grid.unfold(0, size, 1).unfold(1, size, 1)[0:max_x:stride, 0:max_y:stride]      # [len(xs), len(ys), size, size] view
    members = win.reshape(len(xs) * len(ys), size * size)[keep_idx]
    big = torch.iinfo(torch.int64).max
    members = torch.sort(torch.where(members >= 0, members, big), dim=1).values     # (a window's tiles are distinct: no ties to order)
    return torch.where(members == big, -1, members)


def tile_window_lists(members, n_tiles, device, max_votes=None):
    """Invert members [W, 100] (window -> tiles) into int32 [n_tiles, V] (tile -> windows in visiting order, packed,
    -1 padded); V = the largest number of windows any tile belongs to, or `max_votes` when the caller knows a bound (then
    nothing here waits for the device: no size depends on the data)."""
    mem = torch.as_tensor(members, device=device)
    W, S = mem.shape
    win = torch.arange(W, device=device).unsqueeze(1).expand(W, S).reshape(-1)
    flat = mem.reshape(-1)
    tiles = torch.where(flat >= 0, flat, n_tiles)                 # padding goes to a dump row behind the last tile
    tiles, order = torch.sort(tiles, stable=True)                 # row-major input: ascending window id inside every tile
    wins = win[order]
    counts = torch.bincount(tiles, minlength=n_tiles + 1)
    if max_votes is None:
        V = max(int(counts[:n_tiles].max()), 1) if W else 1
    else:
        V = max(int(max_votes), 1)
    start = torch.cumsum(counts, 0) - counts
    rank = torch.arange(tiles.numel(), device=device) - start[tiles]
    rank = torch.where(tiles < n_tiles, rank, 0).clamp_(max=V - 1)
    out = torch.full((n_tiles + 1, V), -1, dtype=torch.int32, device=device)
    out[tiles, rank] = wins.to(torch.int32)
    return out[:n_tiles], counts[:n_tiles]


WINDOW = 10               # tiles per window edge (visualize.py:46-52); every bound below derives from it
HEAD_CHUNK = 4096         # tiles per head product: the fixed grid the one-rank and the sharded run share (same launches -> same bits)


def max_votes_per_tile(stride, size=WINDOW):
    """A tile lies in at most ceil(size / stride)^2 kept windows."""
    return (-(-size // stride)) ** 2


def _shard_info(shard):
    """shard: None | (rank, world) | (rank, world, process_group).  Returns (rank, world, group)."""
    if shard is None:
        return 0, 1, None
    rank, world = int(shard[0]), int(shard[1])
    if not (0 <= rank < world):
        raise ValueError(f"shard rank {rank} outside world {world}")
    if world > 1 and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        raise RuntimeError("window sharding over more than one rank needs an initialised torch.distributed process group")
    return rank, world, (shard[2] if len(shard) > 2 else None)


def window_batch_owner(n_windows, batch_windows, world):
    """The fixed batch grid of a slide's window list and who runs what: batch b = windows [b * batch_windows, ...) goes to rank
    b % world as that rank's slot b // world.  Returns (n_batches, slots_per_rank)."""
    nb = -(-n_windows // batch_windows)
    return nb, -(-nb // world)


@torch.no_grad()
def sliding_window_all_genes_sharded(xtf, ytf, tile_features, model, stride, literal_2d=False, batch_windows=1024, shard=None,
                                     tile_projection=True):
    """visualize.py:35-102 for ONE slide over `world` ranks (BASELINE config 5's multi-GPU form; SURVEY 8e "Config 5": the
    windows of a slide are independent, visualize.py:46-52).  Every rank holds the tile-feature cache and enumerates the
    same window list; then

      1. window batches (the fixed grid of `batch_windows`) are dealt round-robin: rank r runs batches r, r + world, ...
         through the model up to the head's input -- a [batch, D] block per batch;
      2. ONE all-gather of those blocks (W x D fp32: 196 MB at 47 769 windows, D = 1024 -- not the 2 x 4.2 GB an all-reduce
         of per-tile vote sums and counts over 20 820 genes would move, and no cross-rank summation whose order could differ);
      3. tiles are dealt in chunks of HEAD_CHUNK (chunk c to rank c % world): per-tile vote over the gathered window
         vectors in visiting order, then the linear head once per tile.

    In bf16 mode (``tile_projection``, default on) the first layer's local projection runs once per TILE and is gathered per window
    token (ViS.tile_projections / sq_vis_forward_tiles): f is linear in tile feature + position.

    Every launch that touches a window or a tile is the launch the one-rank run makes for it (same batch, same chunk), so
    the rows a rank returns are BIT-IDENTICAL to the one-rank result.  Returns (tile_pred f32 [n_local, G], tile_ids int64
    [n_local] -- the df positions of those rows, ascending --, votes int64 [n_tiles] for the whole slide)."""
    _lib.require_gpu()
    rank, world, group = _shard_info(shard)
    dev = model.flat.device
    mem = enumerate_windows_device(xtf, ytf, stride, dev)           # nothing below waits for the device before the result is read
    feats = tile_features.to(dev, torch.float32).contiguous()
    n_tiles, D = feats.shape
    G = model.cfg.num_outputs
    chunks = range(rank, -(-n_tiles // HEAD_CHUNK), world)
    tile_ids = torch.cat([torch.arange(c * HEAD_CHUNK, min(n_tiles, (c + 1) * HEAD_CHUNK), device=dev) for c in chunks]) if len(chunks) \
        else torch.zeros(0, dtype=torch.int64, device=dev)
    if mem.shape[0] == 0:
        return torch.full((tile_ids.numel(), G), float("nan"), device=dev), tile_ids, torch.zeros(n_tiles, dtype=torch.int64, device=dev)
    gather = (mem[:, 0:1].expand(-1, mem.shape[1]) if literal_2d else mem).to(torch.int32).contiguous()
    # literal_2d: the reference feeds a 2-D [100, D] tensor and takes row 0 -> the prediction depends on the window's
    # first tile only, replicated over the 100 positions
    W = mem.shape[0]
    nb, slots = window_batch_owner(W, batch_windows, world)
    # row of window w in the gathered buffer: [owner rank][slot][offset in batch]
    local = torch.empty(slots * batch_windows, D, dtype=torch.float32, device=dev) if world > 1 else torch.empty(W, D, dtype=torch.float32, device=dev)
    # two batches of windows in flight on two streams (own workspaces): each forward is a chain of dependent
    # launches, a second chain fills the ramp-up / store-drain phases of the first
    main = torch.cuda.current_stream(dev)
    ns = max(1, int(os.environ.get("SQ_SPATIAL_STREAMS", "2")))
    streams = model.__dict__.setdefault("_spatial_streams", None)
    if streams is None or streams[0].device != dev or len(streams) != ns:
        streams = model.__dict__["_spatial_streams"] = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    model._params_lp()                              # refresh the bf16 shadow on the main stream BEFORE the hand-over event:
    # bf16 mode: layer 0's local projection once per TILE (linear in tile feature + position), gathered per window token
    # (sq_vis_forward_tiles) -- n_tiles rows through the product instead of 100 x n_windows
    tile_proj = model.tile_projections(feats) if (tile_projection and model.compute_dtype == _lib.SQ_BF16 and W * mem.shape[1] > n_tiles) else None
    start = torch.cuda.Event()                      # the window streams wait on `start` only and must see the finished cast
    start.record(main)
    for i, b in enumerate(range(rank, nb, world)):
        s = b * batch_windows
        st = streams[i % ns]
        st.wait_event(start)
        with torch.cuda.stream(st):
            o = (b // world) * batch_windows if world > 1 else s
            n = min(batch_windows, W - s)
            local[o:o + n] = model._run_head_inputs(feats, gather[s:s + batch_windows], slot=1 + i % ns, tile_proj=tile_proj)
    for st in streams:
        main.wait_stream(st)
    V = max_votes_per_tile(stride)
    lists, counts = tile_window_lists(mem, n_tiles, dev, max_votes=V)
    if os.environ.get("SQ_SPATIAL_CHECK", "0") not in ("", "0"):        # debugging aid (waits for the device): the vote table's bound
        assert int(counts.max()) <= V, f"a tile lies in {int(counts.max())} windows, the vote table holds {V}"
    if world > 1:
        win_vec = torch.empty(world * slots * batch_windows, D, dtype=torch.float32, device=dev)
        torch.distributed.all_gather(list(win_vec.chunk(world)), local, group=group)
        lists = gathered_row_of_window(lists, batch_windows, world, slots).to(torch.int32)
    else:
        win_vec = local
    if tile_ids.numel() == 0:
        return torch.empty(0, G, dtype=torch.float32, device=dev), tile_ids, counts
    lists = lists[tile_ids].contiguous() if world > 1 else lists.contiguous()
    tile_vec = torch.empty(lists.shape[0], D, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().sq_window_vote(_lib.ptr(win_vec), win_vec.shape[0], D, _lib.ptr(lists), lists.shape[0], lists.shape[1],
                                             1 if stride == WINDOW else 0, float("nan"), _lib.ptr(tile_vec), _lib.stream_ptr(dev)))
    return model.apply_head(tile_vec, chunk=HEAD_CHUNK), tile_ids, counts


@torch.no_grad()
def sliding_window_all_genes(xtf, ytf, tile_features, model, stride, literal_2d=False, batch_windows=1024):
    """All-gene form of visualize.py:35-102 (BASELINE config 5: per-tile 20 820-gene regression): returns
    (tile_pred f32 [n_tiles, G] with NaN for tiles no kept window covers, votes int64 [n_tiles]).

    Nothing of size [windows, 100, D] or [windows, G] exists: a window batch is gathered from the tile-feature cache
    inside the model's first kernel (member indices, -1 = the zero padding of :72-75), the model stops in front of
    its linear head, the per-tile mean / last-writer rule (:87-100) is applied to those D-vectors (sq_window_vote),
    and the head runs ONCE per tile -- mean_w(head(v_w)) = head(mean_w v_w) for a linear head, so HBM sees the
    [n_tiles, G] result once and 100x fewer head products are computed.  (The one-rank case of
    sliding_window_all_genes_sharded.)"""
    out, _, counts = sliding_window_all_genes_sharded(xtf, ytf, tile_features, model, stride, literal_2d=literal_2d,
                                                     batch_windows=batch_windows, shard=None)
    return out, counts


def gathered_row_of_window(w, batch_windows, world, slots):
    """Row of window `w` (tensor or int, -1 = padding stays -1) in the all-gathered buffer [world][slots][batch_windows]."""
    b = w // batch_windows                                          # (-1 stays negative through the floor division)
    row = (b % world) * (slots * batch_windows) + (b // world) * batch_windows + w % batch_windows
    return torch.where(w >= 0, row, w) if torch.is_tensor(w) else (row if w >= 0 else w)


@torch.no_grad()
def sliding_window_method(df, tile_features, model, inds_gene_of_interest, stride, literal_2d=False, batch_windows=512, shard=None):
    """visualize.py:35-102.  df: DataFrame with integer columns xcoord_tf / ycoord_tf (tile grid); tile_features:
    [n_tiles, D] tensor (row i = features of df.iloc[i], i.e. the feature cache); model: ViS on the GPU.
    Returns {gene_index: {tile_index: prediction}} exactly like the reference (stride 10: last writer wins;
    stride < 10: mean over the windows containing the tile).  With ``shard=(rank, world[, group])`` the slide's windows and
    tiles are dealt over the ranks (sliding_window_all_genes_sharded); the requested gene columns of every rank's tiles are
    all-gathered, so every rank returns the same, complete dictionary -- bit-identical to the one-rank call."""
    genes = list(inds_gene_of_interest)
    rank, world, group = _shard_info(shard)
    out, tile_ids, counts = sliding_window_all_genes_sharded(df['xcoord_tf'].values, df['ycoord_tf'].values, tile_features, model, stride,
                                                            literal_2d=literal_2d, batch_windows=batch_windows, shard=shard)
    n_tiles = counts.numel()
    sel_local = out[:, torch.as_tensor(genes, device=out.device)] if genes else out[:, :0]
    if world > 1:
        # equal-sized blocks for the collective: every rank owns at most ceil(chunks / world) head chunks
        per = -(-(-(-n_tiles // HEAD_CHUNK)) // world) * HEAD_CHUNK
        blk = torch.full((per, len(genes) + 1), float("nan"), dtype=torch.float32, device=out.device)
        blk[:tile_ids.numel(), :len(genes)] = sel_local
        blk[:tile_ids.numel(), len(genes)] = tile_ids.to(torch.float32)           # (tile ids < 2^24: exact in fp32)
        blk[tile_ids.numel():, len(genes)] = -1
        allb = torch.empty(world * per, len(genes) + 1, dtype=torch.float32, device=out.device)
        torch.distributed.all_gather(list(allb.chunk(world)), blk, group=group)
        ids = allb[:, len(genes)].to(torch.int64)
        sel_t = torch.full((n_tiles, len(genes)), float("nan"), dtype=torch.float32, device=out.device)
        sel_t[ids[ids >= 0]] = allb[ids >= 0, :len(genes)]
        sel = sel_t.cpu().numpy()
    else:
        sel = sel_local.cpu().numpy() if genes else np.zeros((n_tiles, 0), np.float32)
    cnt = counts.cpu().numpy()
    index = list(df.index)
    preds = {g: {} for g in genes}
    for t in np.nonzero(cnt > 0)[0]:
        for gi, g in enumerate(genes):
            preds[g][index[t]] = sel[t, gi]
    return preds


@torch.no_grad()
def sliding_window_any_model(df, tile_features, model, inds_gene_of_interest, stride, model_type, batch_windows=256):
    """visualize.py:35-102 for the comparator models (``model_type`` 'vit' or 'he2rna'; ViS takes the gather/vote path
    above): window batches [w, 100, D] are built from the feature cache (zero padding, :72-75), HE2RNA gets them as
    channels x tiles (:80-81), and the window predictions of the requested genes are combined per tile with the same
    rule (stride 10: last writer; stride < 10: mean over the windows containing the tile, in visiting order)."""
    genes = list(inds_gene_of_interest)
    members, _ = enumerate_windows(df['xcoord_tf'].values, df['ycoord_tf'].values, stride)
    index = list(df.index)
    preds = {g: {} for g in genes}
    if len(members) == 0 or not genes:
        return preds
    dev = next(model.parameters()).device
    feats = tile_features.to(dev, torch.float32)
    feats_pad = torch.cat([feats, torch.zeros(1, feats.shape[1], device=dev)])           # row -1 = the zero padding
    mem = torch.from_numpy(members).to(dev)
    gsel = torch.as_tensor(genes, device=dev)
    win_pred = []
    for s in range(0, mem.shape[0], batch_windows):
        x = feats_pad[mem[s:s + batch_windows]]                                           # [w, 100, D]
        if model_type == 'he2rna':
            x = x.transpose(1, 2)
        win_pred.append(model(x)[:, gsel].float())
    win_pred = torch.cat(win_pred).cpu().numpy()                                          # [W, len(genes)]
    per_tile = {}
    for w, row in enumerate(members):
        for t in row[row >= 0]:
            if stride == 10:
                per_tile[t] = [w]
            else:
                per_tile.setdefault(t, []).append(w)
    for t, ws in per_tile.items():
        for gi, g in enumerate(genes):
            preds[g][index[t]] = win_pred[ws[0], gi] if stride == 10 else np.mean(win_pred[ws, gi])
    return preds
