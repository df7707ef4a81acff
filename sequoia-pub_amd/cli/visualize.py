"""Spatial prediction maps -- counterpart of /root/reference/spatial_vis/visualize.py:104-307 (same flags and CSV
output ``<out_root>/<project>/<save_folder>/<wsi>/stride-1.csv`` with one column per gene and fold plus the fold mean).

What differs in HOW: every valid tile is embedded ONCE into a feature cache and the windows are gathered from it
(the reference re-reads and re-embeds a tile in every window that holds it); ViS windows run through
``spatial.sliding_window_method`` (gather + vote before the linear head), 'vit' / 'he2rna' through
``spatial.sliding_window_any_model``.  The reference hard-codes its data locations per project; here they are the
defaults of ``--slide_path`` / ``--mask_path`` / ``--checkpoint`` and can be overridden.  Slides are opened with
openslide when it is installed, or given as an ``.npy`` RGB array (``patchgen.ArraySlide``, 20x).  ResNet tiles are
resized to 256 x 256 (the reference's ``Resize((256, 265))`` is not a size the convolution engine takes); UNI tiles to
224 as in the reference."""
import argparse
import os
import pickle

import numpy as np
import pandas as pd
import torch
from scipy.ndimage import binary_dilation

from ..patchgen import ArraySlide
from ..spatial import sliding_window_any_model, sliding_window_method
from .common import init_distributed

BACKGROUND_THRESHOLD = .5


def read_pickle(path):
    """visualize.py:24-32: every object of a (possibly multi-dump) pickle file."""
    objects = []
    with open(path, "rb") as f:
        while True:
            try:
                objects.append(pickle.load(f))
            except EOFError:
                break
    return objects


def valid_tiles(mask, slide_dims, patch_size_resized):
    """visualize.py:181-206: tile origins (level-0 pixels) whose dilated mask window is at least half tissue, and the
    DataFrame with the grid coordinates xcoord_tf / ycoord_tf."""
    downsample_factor = int(slide_dims[0] / mask.shape[0])
    patch_size_in_mask = int(patch_size_resized / downsample_factor)
    mask = np.transpose(mask, axes=[1, 0]) * 1
    valid = []
    for col in range(0, slide_dims[0] - patch_size_resized, patch_size_resized):
        for row in range(0, slide_dims[1] - patch_size_resized, patch_size_resized):
            r, c = int(row / downsample_factor), int(col / downsample_factor)
            win = binary_dilation(mask[r:r + patch_size_in_mask, c:c + patch_size_in_mask], iterations=3)
            if win.sum() >= BACKGROUND_THRESHOLD * win.size:
                valid.append((col, row))
    df = pd.DataFrame(valid, columns=['xcoord', 'ycoord'])
    if len(df):
        df['xcoord_tf'] = ((df['xcoord'] - min(df['xcoord'])) / patch_size_resized).astype(int)
        df['ycoord_tf'] = ((df['ycoord'] - min(df['ycoord'])) / patch_size_resized).astype(int)
    return df


TILE_CHUNK = 512          # tiles per read -> upload -> resize -> embed round: host and device staging stay a few hundred MB


def read_tiles(slide, df, patch_size_resized, lo=0, hi=None):
    """uint8 [hi - lo, p, p, 3]: slide.read_region((col, row), 0, (p, p)) of tiles lo..hi-1 (visualize.py:63-66)."""
    hi = len(df) if hi is None else hi
    tiles = np.empty((hi - lo, patch_size_resized, patch_size_resized, 3), dtype=np.uint8)
    for i, (col, row) in enumerate(zip(df['xcoord'][lo:hi], df['ycoord'][lo:hi])):
        r = slide.read_region((int(col), int(row)), 0, (patch_size_resized, patch_size_resized))
        tiles[i] = np.asarray(r.convert('RGB') if hasattr(r, 'convert') else r)[..., :3]
    return torch.from_numpy(tiles)


def embed_tiles(slide, df, patch_size_resized, out_size, feat_model, device, chunk=None, shard=None):
    """Feature cache [n_tiles, D] on the device: tiles go through in chunks -- read, upload, resize ON THE DEVICE
    (antialiased bilinear, uni.resize_u8) when the read size differs from the extractor's input, embed -- so only the
    features stay resident (a 40x slide with 50 000 valid tiles of 512 x 512 would otherwise need ~40 GB of host
    uint8 plus the fp32 resize copies).  With ``shard=(rank, world[, group])`` the chunks are dealt round-robin over the
    ranks (chunk c to rank c % world: the same launches the one-rank run makes for those tiles) and the cache is
    all-gathered once (SURVEY 8e "Config 5": n_tiles x D fp32), so every rank ends with the complete, identical cache."""
    from ..spatial import _shard_info, gathered_row_of_window
    from ..uni import resize_u8
    rank, world, group = _shard_info(shard)
    chunk = int(chunk or TILE_CHUNK)
    D = 2048 if hasattr(feat_model, 'conv1') else 1024
    feats = []
    starts = list(range(0, len(df), chunk))
    for lo in starts[rank::world]:
        t = read_tiles(slide, df, patch_size_resized, lo, min(lo + chunk, len(df))).to(device)
        if patch_size_resized != out_size:
            t = resize_u8(t, out_size)
        feats.append(feat_model.extract_patches_u8(t))
    if world == 1:
        if not feats:
            return torch.empty(0, D, device=device)      # what the extractor would have returned for 0 tiles
        return torch.cat(feats, 0)
    slots = -(-len(starts) // world)
    local = torch.zeros(slots * chunk, D, dtype=torch.float32, device=device)
    for i, f in enumerate(feats):
        local[i * chunk:i * chunk + f.shape[0]] = f
    allf = torch.empty(world * slots * chunk, D, dtype=torch.float32, device=device)
    torch.distributed.all_gather(list(allf.chunk(world)), local, group=group)
    return allf[gathered_row_of_window(torch.arange(len(df), device=device), chunk, world, slots)]


def open_slide(path):
    if path.endswith('.npy'):
        return ArraySlide([np.load(path)])
    import openslide                                   # absent from this image; real .svs / .tif need it
    return openslide.OpenSlide(path)


def build_model(model_type, input_dim, n_genes, device, compute_dtype):
    if model_type == 'vis':
        from ..vis import ViS
        return ViS(num_outputs=n_genes, input_dim=input_dim, depth=6, nheads=16, dimensions_f=64, dimensions_c=64, dimensions_s=64,
                   device=str(device), compute_dtype=compute_dtype)
    if model_type == 'vit':
        from ..vit import ViT
        return ViT(num_outputs=n_genes, dim=input_dim, depth=6, heads=16, mlp_dim=2048, dim_head=64, device=str(device), compute_dtype=compute_dtype)
    from ..he2rna import HE2RNA
    return HE2RNA(input_dim=input_dim, layers=[256, 256], ks=[1, 2, 5, 10, 20, 50, 100], output_dim=n_genes, device=str(device))


def main(argv=None):
    p = argparse.ArgumentParser(description='Getting features')
    p.add_argument('--study', type=str, help='cancer study abbreviation, lowercase')
    p.add_argument('--project', type=str, help='name of project (spatial_GBM_pred, TCGA-GBM, PESO, Breast-ST)')
    p.add_argument('--gene_names', type=str, help='genes to visualize, separated by commas, a .npy list, or "all"')
    p.add_argument('--wsi_file_name', type=str, help='wsi filename')
    p.add_argument('--save_folder', type=str, help='destination folder')
    p.add_argument('--model_type', type=str, help='model to use:  "he2rna", "vit" or "vis"')
    p.add_argument('--feat_type', type=str, help='"resnet" or "uni"')
    p.add_argument('--folds', type=str, default='0,1,2,3,4', help='folds to use in prediction split by comma')
    p.add_argument('--slide_path', type=str, default=None, help='directory of the slide (default: ./TCGA/<project>/)')
    p.add_argument('--mask_path', type=str, default=None, help='mask .npy (default: ./TCGA/<project>_Masks/<wsi>/mask.npy)')
    p.add_argument('--checkpoint', type=str, default=None, help='fold checkpoints + test_results.pkl (default: <model_type>_<feat_type>/<study>/)')
    p.add_argument('--out_root', type=str, default='./visualizations')
    p.add_argument('--extractor_weights', type=str, default=None, help='resnet50 / UNI state dict (default: torchvision url / ./uni_ckpt/pytorch_model.bin)')
    p.add_argument('--resize_factor', type=float, default=None, help='level-0 pixels per 20x pixel (default: aperio.AppMag / 20)')
    p.add_argument('--tile_chunk', type=int, default=TILE_CHUNK, help='tiles per read -> upload -> embed round (and the unit dealt over the ranks under torchrun)')
    p.add_argument('--compute_dtype', default='bf16', choices=['fp32', 'bf16', 'f16x3', 'bf16x3'],
                   help='f16x3 / bf16x3: the ResNet-50 extractor on split planes (fp32-class features); the aggregator then runs in fp32')
    args = p.parse_args(argv)
    assert args.feat_type in ['resnet', 'uni'] and args.model_type in ['vit', 'vis', 'he2rna']
    # under torchrun: ONE slide over the ranks -- tile chunks for the feature cache, window batches and tile chunks for the
    # aggregator (spatial.sliding_window_all_genes_sharded); rank 0 writes the CSV.  Alone: cuda:0 as in the reference.
    rank, world, device = init_distributed()
    shard = (rank, world) if world > 1 else None
    stride, patch_size = 1, 256                         # 256 px at 20x (0.5 um / px)

    checkpoint = args.checkpoint or f'{args.model_type}_{args.feat_type}/{args.study}/'
    gene_ids = list(read_pickle(os.path.join(checkpoint, 'test_results.pkl'))[0]['genes'])
    save_path = os.path.join(args.out_root, args.project, args.save_folder, args.wsi_file_name)
    if rank == 0:
        os.makedirs(save_path, exist_ok=True)
    if args.gene_names != 'all':
        gene_names = list(np.load(args.gene_names, allow_pickle=True)) if '.npy' in args.gene_names else args.gene_names.split(",")
    else:
        gene_names = gene_ids

    stem = args.wsi_file_name.replace('.svs', '').replace('.tif', '').replace('.npy', '')
    slide_dir = args.slide_path or f'./TCGA/{args.project}/'
    mask = np.load(args.mask_path or f'./TCGA/{args.project}_Masks/{stem}/mask.npy')
    slide = open_slide(os.path.join(slide_dir, args.wsi_file_name))
    resize_factor = args.resize_factor if args.resize_factor is not None else float(slide.properties.get('aperio.AppMag', 20)) / 20.0
    patch_size_resized = int(resize_factor * patch_size)
    df = valid_tiles(mask, slide.dimensions, patch_size_resized)
    print('Got dataframe of valid tiles')

    # ---- feature cache: every valid tile embedded once
    input_dim = 2048 if args.feat_type == 'resnet' else 1024
    split = args.compute_dtype in ('f16x3', 'bf16x3')          # the ResNet-50 extractor's modes; everything else then runs in fp32
    if split and args.feat_type != 'resnet':
        raise SystemExit('--compute_dtype f16x3 / bf16x3 are the ResNet-50 extractor\'s modes; --feat_type uni runs in fp32 or bf16')
    agg_dtype = 'fp32' if split else args.compute_dtype
    if args.feat_type == 'resnet':
        from ..resnet import resnet50
        feat_model = resnet50(pretrained=args.extractor_weights is None, compute_dtype=args.compute_dtype)
        if args.extractor_weights:
            feat_model.load_state_dict(torch.load(args.extractor_weights, map_location='cpu'))
        feat_model = feat_model.to(device).eval()
        tile_features = embed_tiles(slide, df, patch_size_resized, 256, feat_model, device, chunk=args.tile_chunk, shard=shard)
    else:
        from ..uni import create_model
        feat_model = create_model("vit_large_patch16_224", img_size=224, patch_size=16, init_values=1e-5, num_classes=0,
                                  dynamic_img_size=True, compute_dtype=args.compute_dtype)
        feat_model.load_state_dict(torch.load(args.extractor_weights or "./uni_ckpt/pytorch_model.bin", map_location='cpu'), strict=True)
        feat_model = feat_model.to(device).eval()
        tile_features = embed_tiles(slide, df, patch_size_resized, 224, feat_model, device, chunk=args.tile_chunk, shard=shard)

    # ---- fold ensemble (visualize.py:248-300)
    res_df = df.copy(deep=True)
    folds = [int(i) for i in args.folds.split(',')]
    inds = []
    for g in gene_names:
        if g in gene_ids:
            inds.append(gene_ids.index(g))
        else:
            print('gene not in predicted values ' + str(g))
    for fold in folds:
        fold_ckpt = os.path.join(checkpoint, 'model_best_' + str(fold) + '.pt')
        if fold == 0 and args.model_type in ('vit', 'vis'):
            fold_ckpt = fold_ckpt.replace('_0', '')
        model = build_model(args.model_type, input_dim, len(gene_ids), device, agg_dtype)
        if args.model_type == 'he2rna':
            obj = torch.load(fold_ckpt.replace('best_', ''), map_location='cpu', weights_only=False)      # he2rna.fit pickles the model
            model.load_state_dict(obj.state_dict() if hasattr(obj, 'state_dict') else obj)
        else:
            model.load_state_dict(torch.load(fold_ckpt, map_location='cpu'))
        model = model.to(device).eval()
        if args.model_type == 'vis':
            preds = sliding_window_method(df, tile_features, model, inds, stride, shard=shard)
        else:
            preds = sliding_window_any_model(df, tile_features, model, inds, stride, args.model_type)
        for ind_gene in inds:
            res_df[gene_ids[ind_gene] + '_' + str(fold)] = res_df.index.map(preds[ind_gene])
    for ind_gene in inds:
        res_df[gene_ids[ind_gene]] = res_df[[gene_ids[ind_gene] + '_' + str(i) for i in folds]].mean(axis=1)
    save_name = os.path.join(save_path, 'stride-' + str(stride) + '.csv')
    if rank == 0:
        res_df.to_csv(save_name)
    if world > 1:
        torch.distributed.barrier()
    print('Done')
    return res_df, save_name


if __name__ == '__main__':
    main()
