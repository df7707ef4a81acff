"""WSI -> patch HDF5 files -- counterpart of /root/reference/pre_processing/patch_gen_hdf5.py:148-208 (same flags,
same outputs: ``<patch_path>/<slide>/<slide>.hdf5`` + ``complete.txt``, ``<mask_path>/<slide>/mask.npy``).  Reading
real ``.svs`` / ``.tiff`` slides needs openslide-python, exactly like the reference; the mask / tiling / filter logic is
sequoia-pub_amd/patchgen.py."""
import argparse
import os
from multiprocessing import Pool

import pandas as pd

from ..patchgen import extract_patches


def get_slide_id(slide_name):
    return slide_name.split('.')[0]


def process(opts):
    slide_path, patch_size, patches_output_dir, mask_path, slide_id, max_patches_per_slide = opts
    try:
        from openslide import OpenSlide
    except ImportError as e:
        raise SystemExit("openslide-python is required to read whole-slide images (pip install openslide-python); "
                         "sequoia-pub_amd.patchgen.extract_patches also accepts any object with OpenSlide's interface") from e
    extract_patches(OpenSlide(slide_path), mask_path, patch_size, patches_output_dir, slide_id, max_patches_per_slide)


def main(argv=None):
    p = argparse.ArgumentParser(description='Generate patches from a given folder of images')
    p.add_argument('--ref_file', default="examples/ref_file.csv", required=False, metavar='ref_file', type=str)
    p.add_argument('--wsi_path', default="examples/HE", metavar='WSI_PATH', type=str)
    p.add_argument('--patch_path', default="examples/Patches_hdf5", metavar='PATCH_PATH', type=str)
    p.add_argument('--mask_path', default="examples/Patches_hdf5", metavar='MASK_PATH', type=str)
    p.add_argument('--patch_size', default=256, type=int)
    p.add_argument('--start', type=int, default=0)
    p.add_argument('--end', type=int, default=None)
    p.add_argument('--max_patches_per_slide', default=None, type=int)
    p.add_argument('--debug', default=0, type=int)
    p.add_argument('--parallel', default=1, type=int)
    args = p.parse_args(argv)

    slide_list = [s for s in os.listdir(args.wsi_path) if s.endswith('.svs') or s.endswith('.tiff')]
    if args.ref_file:
        wanted = {f'{s}.svs' for s in pd.read_csv(args.ref_file)['wsi_file_name']}
        slide_list = sorted(set(slide_list) & wanted)
    slide_list = slide_list[args.start:args.end] if args.end is not None else slide_list[args.start:]
    if args.debug:
        slide_list = slide_list[0:5]
        args.max_patches_per_slide = 20
    print(f"Found {len(slide_list)} slides")
    opts = [(os.path.join(args.wsi_path, s), (args.patch_size, args.patch_size), args.patch_path, args.mask_path,
             get_slide_id(s), args.max_patches_per_slide) for s in slide_list]
    if args.parallel:
        with Pool(processes=4) as pool:
            pool.map(process, opts)
    else:
        for o in opts:
            process(o)


if __name__ == '__main__':
    main()
