"""Sliding-window spatial inference -- host-side mirror of /root/reference/spatial_vis/visualize.py:35-102
(``sliding_window_method``): 10x10-tile windows over the tile grid at a given stride, kept if they hold more
than 50 tiles, zero-padded to 100 tokens, one model forward per window, the window's prediction written to (or
averaged over, for stride < 10) every tile of the window.

Differences in HOW (not what): every tile is embedded ONCE into a feature cache (the reference re-embeds a tile
in every window that contains it, up to 100 times); windows are enumerated with tensor ops and run through the
model in large batches (BASELINE config 5: ~47k windows of 100 tokens).  ``literal_2d=True`` reproduces the
reference's 2-D input quirk (the prediction depends only on the window's first tile)."""
import numpy as np
import torch


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


@torch.no_grad()
def sliding_window_method(df, tile_features, model, inds_gene_of_interest, stride, literal_2d=False, batch_windows=512):
    """visualize.py:35-102.  df: DataFrame with integer columns xcoord_tf / ycoord_tf (tile grid); tile_features:
    [n_tiles, D] tensor (row i = features of df.iloc[i], i.e. the feature cache); model: ViS on the GPU.
    Returns {gene_index: {tile_index: prediction}} exactly like the reference (stride 10: last writer wins;
    stride < 10: mean over the windows containing the tile)."""
    members, _ = enumerate_windows(df['xcoord_tf'].values, df['ycoord_tf'].values, stride)
    genes = list(inds_gene_of_interest)
    preds = {g: {} for g in genes}
    if len(members) == 0:
        return preds
    dev = model.flat.device
    feats = tile_features.to(dev, torch.float32)
    D = feats.shape[1]
    feats_pad = torch.cat([feats, torch.zeros(1, D, device=dev)])               # index -1 -> zero row (padding)
    gidx = torch.as_tensor(genes, device=dev)
    n_tiles = feats.shape[0]
    acc = torch.zeros(n_tiles, len(genes), dtype=torch.float64, device=dev)
    cnt = torch.zeros(n_tiles, dtype=torch.float64, device=dev)
    last = torch.zeros(n_tiles, len(genes), dtype=torch.float32, device=dev)
    mem = torch.from_numpy(members).to(dev)
    for s in range(0, len(members), batch_windows):
        m = mem[s:s + batch_windows]                                            # [w, 100]
        x = feats_pad[m]                                                        # [w, 100, D] (-1 indexes the zero row)
        if literal_2d:
            # reference: model(features_all) with a 2-D [100, D] tensor, then [0]  -> depends on tile 0 only
            out = model(x[:, 0:1, :].expand(-1, 100, -1).contiguous())
        else:
            out = model(x)
        out = out[:, gidx]                                                      # [w, n_genes]
        valid = m >= 0
        rows = m[valid]
        vals = out.unsqueeze(1).expand(-1, m.shape[1], -1)[valid]               # [n_valid, n_genes]
        acc.index_add_(0, rows, vals.double())
        cnt.index_add_(0, rows, torch.ones_like(rows, dtype=torch.float64))
        last[rows] = vals            # windows are visited in the reference's (x, y) order; later batches overwrite
    cnt_c, acc_c, last_c = cnt.cpu().numpy(), acc.cpu().numpy(), last.cpu().numpy()
    index = list(df.index)
    for t in np.nonzero(cnt_c > 0)[0]:
        for gi, g in enumerate(genes):
            preds[g][index[t]] = last_c[t, gi] if stride == 10 else np.float32(acc_c[t, gi] / cnt_c[t])
    return preds
