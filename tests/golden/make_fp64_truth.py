"""The reference's ResNet-50 (src/resnet.py) in DOUBLE precision on the probe patches of the four golden slides -> fp64_truth.npz.

Why: the goldens hold the reference's fp32 features.  How far is fp32 arithmetic itself from the exact value of the same network?
On the He-init weight set ~5e-7 of the largest feature; on the wide-range set (BN gamma 0.1 .. 10, folded scales over four
decades) ~1e-5 -- that network amplifies every rounding 20 times more (a 2^-24 relative perturbation of the INPUT alone moves its
features by 1.6e-6).  A distance of 1.65e-5 between the split-fp16 mode and the fp32 golden on that slide (round-5 review, weak
#2) is therefore a distance between two roundings of the same number, not an accuracy defect of one of them; with the exact
features on file the GPU tests can hold every mode to "as close to the truth as the reference's own fp32 is" (tests/test_gpu_pipeline.py).

Run in the build container only (imports /root/reference):   python tests/golden/make_fp64_truth.py      (~10 minutes on 8 cores)
Writes data only: per slide the probe row indices, the fp64 features of those patches and the fp32 golden's distance from them."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from src.resnet import resnet50                      # noqa: E402  (reference)
from oracle import resnet_oracle                     # noqa: E402
import sequoia_pub_amd                               # noqa: E402,F401
from sequoia_pub_amd import synth                    # noqa: E402

torch.set_num_threads(8)
MAX_PROBES = 16          # rows of feat_probe used per slide (every k-th, so that the whole slide is sampled)

SLIDES = {        # fixture, patches, weight set -- the table of bench.py GOLDEN_SLIDES
    "noise224": ("pipeline_slide.npz", lambda: synth.patches_u8(7, 1000, 224), "std"),
    "struct224": ("pipeline_slide_struct224.npz", lambda: synth.structured_patches_u8(11, 1000, 224), "std"),
    "struct256": ("pipeline_slide_struct256.npz", lambda: synth.structured_patches_u8(12, 1000, 256), "std"),
    "wide224": ("pipeline_slide_wide224.npz", lambda: synth.structured_patches_u8(13, 1000, 224), "wide"),
}


def main():
    out = {}
    nets = {}
    for key, (fixture, make, weights) in SLIDES.items():
        z = np.load(os.path.join(HERE, fixture))
        step = int(z["probe_step"]) if "probe_step" in z else 64
        n_probe = z["feat_probe"].shape[0]
        sel = np.arange(0, n_probe, max(1, n_probe // MAX_PROBES))[:MAX_PROBES]           # rows of feat_probe
        rows = sel * step                                                                  # patch indices in the slide
        if weights not in nets:
            sd = resnet_oracle.init_resnet50_state_dict(seed=99, perturb_bn=True) if weights == "std" else \
                resnet_oracle.init_resnet50_state_dict_wide(123, running_stats=np.load(os.path.join(HERE, "resnet50_wide_bn.npz")))
            rn = resnet50(pretrained=False)
            full = rn.state_dict()
            full.update(sd)
            rn.load_state_dict(full)
            nets[weights] = rn.double().eval()
        patches = make()[rows]
        x = resnet_oracle.transform_patch_u8(patches).double()       # the fp32 transform of the reference, then exact
        with torch.no_grad():
            f64 = torch.cat([nets[weights].forward_extract(x[i:i + 4]) for i in range(0, len(x), 4)]).numpy()
        f32 = z["feat_probe"][sel].astype(np.float64)
        dist = float(np.abs(f32 - f64).max() / np.abs(f64).max())
        out[key + "_probe_rows"] = sel.astype(np.int64)
        out[key + "_patch_rows"] = rows.astype(np.int64)
        out[key + "_features_fp64"] = f64
        out[key + "_fp32_golden_rel_dist"] = np.array(dist)
        print(f"{key}: {len(rows)} patches, reference fp32 golden vs the same network in fp64: {dist:.3e} of max |feature| {np.abs(f64).max():.3f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "fp64_truth.npz"), **out)


if __name__ == "__main__":
    main()
