"""Patch features -- counterpart of /root/reference/pre_processing/compute_features_hdf5.py (same flags,
same files: reads ``<patch_data_path>/<slide>/<slide>.hdf5`` (one uint8 [S,S,3] dataset per tile), writes
``<feature_path>/<project>/<WSI>/<WSI>.h5`` dataset ``"{feat_type}_features"`` [n, 2048] fp32 and
``complete_tile.txt``).  The per-patch batch-1 loop with two PCIe syncs per patch (:116-123) becomes one
upload of the slide's patches and batched HIP launches with the transform fused: ResNet-50 (``--feat_type resnet``,
[n, 2048]) or the UNI ViT-L/16 (``--feat_type uni``, [n, 1024]; ``--weights`` = UNI's ``pytorch_model.bin``).

    python -m sequoia_pub_amd.cli.compute_features --feat_type resnet --ref_file ref.csv \
        --patch_data_path Patches_hdf5 --feature_path features [--start i --end j]
Under torchrun the slide list is sharded across the GPUs (the reference's --start/--end, automatically)."""
import argparse
import os
import random

import numpy as np
import torch

from .. import store
from ..data import shard_rows
from ..resnet import resnet50
from .common import init_distributed, ref_frame, seed_everything


def main(argv=None):
    parser = argparse.ArgumentParser(description='Getting features')
    parser.add_argument('--feat_type', default="resnet", type=str, required=True, help='"resnet" or "uni"')
    parser.add_argument('--ref_file', type=str, required=True, help='Path with reference csv file')
    parser.add_argument('--patch_data_path', type=str, required=True, help='Directory where the patch is saved')
    parser.add_argument('--feature_path', type=str, default="/examples/features", help='Output directory to save features')
    parser.add_argument('--max_patch_number', type=int, default=4000, help='Max number of patches to use per slide')
    parser.add_argument('--seed', type=int, default=99, help='Seed for random generation')
    parser.add_argument("--tcga_projects", help="the tcga_projects we want to use", default=None, type=str, nargs='*')
    parser.add_argument('--start', type=int, default=0, help='Start slide index for parallelization')
    parser.add_argument('--end', type=int, default=None, help='End slide index for parallelization')
    parser.add_argument('--compute_dtype', default='fp32', choices=['fp32', 'bf16', 'f16x3', 'bf16x3'],
                        help='resnet: f16x3 = split-fp16 planes, fp32-class results at 2.5x the exact fp32 mode (what bench.py quotes); '
                             'bf16 = throughput mode (not label-exact); uni: fp32 or bf16')
    parser.add_argument('--weights', type=str, default=None,
                        help='resnet: torchvision resnet50 state_dict (.pth), default model_zoo URL; uni: path of pytorch_model.bin (required)')
    args = parser.parse_args(argv)
    seed_everything(args.seed)
    rank, world, device = init_distributed()
    if args.feat_type == 'resnet':
        model = resnet50(pretrained=args.weights is None, compute_dtype=args.compute_dtype)
        if args.weights:
            model.load_state_dict(torch.load(args.weights, map_location='cpu'))
    elif args.feat_type == 'uni':                                            # compute_features_hdf5.py:62-68
        if args.compute_dtype not in ('fp32', 'bf16'):
            raise SystemExit('--feat_type uni runs in fp32 or bf16 (the split modes are the ResNet-50 embedder\'s)')
        from ..uni import create_model
        model = create_model("vit_large_patch16_224", img_size=224, patch_size=16, init_values=1e-5, num_classes=0,
                             dynamic_img_size=True, compute_dtype=args.compute_dtype)
        if args.weights:
            model.load_state_dict(torch.load(args.weights, map_location='cpu'), strict=True)
        elif os.environ.get("SEQUOIA_ALLOW_RANDOM_UNI") != "1":
            raise SystemExit('--feat_type uni needs --weights <path to UNI pytorch_model.bin> (gated download, hf.co/MahmoodLab/UNI)')
    else:
        raise SystemExit('please specify feat_type "resnet" or "uni"')
    model.to(device).eval()
    df = ref_frame(args.ref_file, args.tcga_projects, args.start, args.end)
    lo, hi = shard_rows(df.shape[0], rank, world)
    df = df.iloc[lo:hi]
    print(f'Number of slides = {df.shape[0]} (rank {rank}/{world})')
    for _, row in df.iterrows():
        WSI = row['wsi_file_name']
        WSI_slide = WSI.split('.')[0]
        project = row['tcga_project']
        WSI = WSI.replace('.svs', '')
        if not os.path.exists(os.path.join(args.patch_data_path, WSI_slide)):
            print('Not exist {}'.format(os.path.join(args.patch_data_path, WSI_slide)))
            continue
        path = os.path.join(args.patch_data_path, WSI_slide, WSI_slide + '.hdf5')
        path_h5 = os.path.join(args.feature_path, project, WSI)
        os.makedirs(path_h5, exist_ok=True)
        if os.path.exists(os.path.join(path_h5, "complete_resnet.txt")):     # (sic) the reference checks this name
            print(f'{WSI}: Resnet features already obtained')
            continue
        try:
            with store.File(path, 'r') as f_read:
                keys = list(f_read.keys())
                if len(keys) > args.max_patch_number:
                    keys = random.sample(keys, args.max_patch_number)
                patches = np.stack([np.asarray(f_read[key][:]) for key in keys])
            patches = torch.from_numpy(patches)
            if args.feat_type == 'uni' and patches.shape[1] != 224:              # transforms.Resize(224), :54
                from ..uni import resize_u8
                patches = resize_u8(patches.to(device), 224)
            feats = model.extract_patches_u8(patches, sub_batch=1000).cpu().numpy()      # clamped per mode / patch size by the extractor
            f_write = store.File(os.path.join(path_h5, WSI + '.h5'), "w")
            f_write.create_dataset(f"{args.feat_type}_features", data=feats)
            f_write.close()
            with open(os.path.join(path_h5, "complete_tile.txt"), 'w') as f_sum:
                f_sum.write(f"Total n patch = {len(feats)}")
        except Exception as e:                                               # :141-144 skip the slide, keep going
            print(e)
            print(WSI)
            continue


if __name__ == '__main__':
    main()
